"""Kernel names as the summaries key them: `void k_partition_stats<SigF32>(SigF32, int, ...)` -> `k_partition_stats`,
`void k_n1_hist<1, SigI16>(...)` -> `k_n1_hist<1>`: the signal-matrix template argument (float32 / int16 rows) is dropped -- the
workload a launch belongs to says which it was -- the numeric ones stay."""
import re


def kname(name: str) -> str:
    k = name.split("(")[0].replace("void ", "")
    k = re.sub(r"Sig(F32|I16)\s*,\s*", "", k)
    k = re.sub(r",?\s*Sig(F32|I16)\b", "", k)
    return k.replace("<>", "")
