"""Command line + JSON config with the reference's flags (torchrl/utils/args.py:6-53), table-driven."""
import argparse
import json

import torch

# (flag, type or None for a switch, default, help)
_FLAGS = (
    ("seed", int, 0, "random seed"),
    ("vec_env_nums", int, 4, "vec env nums"),
    ("proc_nums", int, 4, "vec env process nums"),
    ("eval_worker_nums", int, 2, "eval worker nums"),
    ("config", str, None, "config file"),
    ("save_dir", str, "./snapshots", "directory for snapshots"),
    ("log_dir", str, "./log", "directory for logs"),
    ("no_cuda", None, False, "disables GPU training"),
    ("overwrite", None, False, "overwrite previous experiments"),
    ("device", int, 0, "gpu specification"),
    ("id", str, None, "experiment id"),
)


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="RL")
    for name, kind, default, text in _FLAGS:
        if kind is None:
            parser.add_argument("--" + name, action="store_true", default=default, help=text)
        else:
            parser.add_argument("--" + name, type=kind, default=default, help=text)
    args = parser.parse_args(argv)
    args.cuda = torch.cuda.is_available() and not args.no_cuda
    return args


def get_params(file_name):
    with open(file_name) as handle:
        return json.load(handle)
