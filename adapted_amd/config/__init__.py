from .sig_proc import (SigProcConfig, get_chemistry_specific_config, get_config,  # noqa: F401
                       load_nested_config_from_file, nested_config_from_dict)
