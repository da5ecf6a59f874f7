"""Kernel names as the summaries key them: `void k_partition_stats<SigF32>(SigF32, int, ...)` -> `k_partition_stats`,
`void k_n1_hist<1, SigI16>(...)` -> `k_n1_hist<1>`: the signal-matrix template argument (float32 / int16 rows) is dropped -- the
workload a launch belongs to says which it was -- the numeric ones stay."""
import re


def _mangled(name: str):
    """`_Z13k_cnn_conv64sILi3ELb0EEvPKDF16_...` -> `k_cnn_conv64s<3, 0>`: rocprofv3 leaves kernels with _Float16 parameters mangled"""
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return None
    n = int(m.group(1))
    ident, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while True:
            a = re.match(r"L[ibjlm](n?\d+)E", rest)
            if not a:
                break
            args.append(a.group(1).replace("n", "-"))
            rest = rest[a.end():]
    return ident + ("<%s>" % ", ".join(args) if args else "")


def kname(name: str) -> str:
    if name.startswith("_Z"):
        k = _mangled(name)
        if k:
            return k
    k = name.split("(")[0].replace("void ", "")
    k = re.sub(r"Sig(F32|I16)\s*,\s*", "", k)
    k = re.sub(r",?\s*Sig(F32|I16)\b", "", k)
    return k.replace("<>", "")
