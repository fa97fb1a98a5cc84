"""Logger with the reference's interface (torchrl/utils/logger.py:17-158):
`add_update_info(dict)` accumulates per-update scalars, `add_epoch_info(...)`
emits mean/std/max/min of each over the epoch plus the epoch scalars to stdout
(tabulate if installed) and `log.csv`; tensorboardX / wandb / git are used only
if importable (the reference hard-requires them and `params['project']`)."""
import csv
import json
import logging
import os
import shutil
import sys

import numpy as np

try:
    from tabulate import tabulate
except Exception:                                           # pragma: no cover
    tabulate = None
try:
    import tensorboardX
except Exception:
    tensorboardX = None
try:
    import wandb
except Exception:
    wandb = None


def _jsonable(obj):
    try:
        json.dumps(obj)
        return obj
    except TypeError:
        if isinstance(obj, dict):
            return {k: _jsonable(v) for k, v in obj.items()}
        return repr(obj)


class Logger:
    def __init__(self, experiment_id, env_name, seed, params, log_dir="./log", overwrite=False):
        self.logger = logging.getLogger("{}_{}_{}".format(experiment_id, env_name, str(seed)))
        self.logger.handlers = []
        self.logger.propagate = False
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter("%(asctime)s %(threadName)s %(levelname)s: %(message)s"))
        handler.setLevel(logging.INFO)
        self.logger.addHandler(handler)
        self.logger.setLevel(logging.INFO)

        self.work_dir = os.path.join(log_dir, experiment_id, env_name, str(seed))
        # one process per GPU: only rank 0 touches the log directory (every rank holds the same global statistics)
        # -- the rank of the torch.distributed group this package shards over, not the raw RANK variable: an independent
        # run started under a launcher (a seed sweep under torchrun / SLURM) is no shard and keeps its logs and snapshots
        from .. import dist
        self.is_writer = dist.rank() == 0
        if not dist.initialized() and os.environ.get("RANK", "0") not in ("", "0"):
            self.logger.warning("RANK=%s is set but no torch.distributed process group exists: this process writes its "
                                "own logs and snapshots (initialise the group before building the Logger to shard)",
                                os.environ["RANK"])
        if self.is_writer:
            if os.path.exists(self.work_dir):
                assert overwrite, "Experiment Exists and Did not set overwrite"
                shutil.rmtree(self.work_dir)
            os.makedirs(self.work_dir, exist_ok=True)
        self.tf_writer = tensorboardX.SummaryWriter(self.work_dir) if (tensorboardX is not None and self.is_writer) else None
        self.csv_file_path = os.path.join(self.work_dir, 'log.csv')
        self.update_count = 0
        self.stored_infos = {}
        self._later = []
        if self.is_writer:
            with open(os.path.join(self.work_dir, 'params.json'), 'w') as f:
                json.dump(_jsonable(params), f, indent=2)
        self.logger.info("Experiment Name:{}".format(experiment_id))
        params["name_combine"] = "{}_{}".format(experiment_id, env_name)
        self.use_wb = wandb is not None and params.get('project') is not None and self.is_writer
        if self.use_wb:
            wandb.init(project=params['project'], name="{}_{}_{}".format(experiment_id, env_name, str(seed)),
                       group="{}_{}".format(experiment_id, env_name), config=_jsonable(params))

    def finish(self):
        if self.use_wb:
            wandb.finish()
        if self.tf_writer is not None:
            self.tf_writer.close()

    def log(self, info):
        self.logger.info(info)

    def add_update_info(self, infos):
        self._drain()                                                    # (keeps the order of arrival)
        for key, value in infos.items():
            self.stored_infos.setdefault(key, []).append(value)
        self.update_count += 1

    def add_update_infos_later(self, resolve):
        """Not in the reference: `resolve()` returns the info dicts of updates that have been launched on the device but
        not waited for; they are taken (in order) when the next row is written or the next dict arrives.  The epoch
        loop can then launch the next rollout before the update's statistics have come back."""
        self._later.append(resolve)
        if len(self._later) >= 64:                                       # a row is a long way off: do not let the launched
            self._drain()                                                # updates' device-side statistics pile up

    def _drain(self):
        later, self._later = self._later, []
        for resolve in later:
            for infos in resolve():
                for key, value in infos.items():
                    self.stored_infos.setdefault(key, []).append(value)
                self.update_count += 1

    def _update_statistics(self):
        """[(name, {Mean, Std, Max, Min})] of every scalar logged by add_update_info since the last epoch row."""
        self._drain()
        out = []
        for key, values in self.stored_infos.items():
            arr = np.asarray(values, dtype=np.float64)
            out.append((key, {"Mean": arr.mean(), "Std": arr.std(), "Max": arr.max(), "Min": arr.min()}))
        return out

    def _print_tables(self, epoch_scalars, statistics):
        table = [[key, "{:.5f}".format(float(value))] for key, value in epoch_scalars.items()]
        stat_table = [[key] + ["{:.5f}".format(v) for v in stats.values()] for key, stats in statistics]
        if tabulate is None:
            for line in table + stat_table:
                self.logger.info(" ".join(str(x) for x in line))
            return
        self.logger.info("\n" + tabulate(table))
        if stat_table:
            self.logger.info("\n" + tabulate(stat_table, ["Name", "Mean", "Std", "Max", "Min"]))

    def add_epoch_info(self, epoch_num, total_frames, total_time, infos, csv_write=True):
        """One row per epoch: the epoch scalars in the order given, then Mean / Std / Max / Min of every per-update
        scalar; to stdout, tensorboard / wandb when present, and `log.csv` (header written with epoch 0)."""
        for label, value in (("EPOCH:{}", epoch_num), ("Time Consumed:{}s", total_time), ("Total Frames:{}s", total_frames)):
            self.logger.info(label.format(value))
        statistics = self._update_statistics()
        columns = [("EPOCH", epoch_num), ("Time Consumed", total_time), ("Total Frames", total_frames)]
        columns += list(infos.items())
        columns += [("{}_{}".format(key, name), v) for key, stats in statistics for name, v in stats.items()]
        scalars = dict(columns[3:])
        if self.tf_writer is not None:
            for key, v in scalars.items():
                self.tf_writer.add_scalar(key, v, total_frames)
        if self.use_wb:
            wandb.log(scalars, step=total_frames)
        self._print_tables(dict(infos), statistics)
        if csv_write and self.is_writer:                                 # values formatted as the reference writes them
            with open(self.csv_file_path, 'a') as handle:                  # (utils/logger.py:108-130: '{:.5f}' after the
                writer = csv.writer(handle)                                #  three leading columns)
                if epoch_num == 0:
                    writer.writerow([name for name, _ in columns])
                writer.writerow([value for _, value in columns[:3]] + ["{:.5f}".format(float(v)) for _, v in columns[3:]])
        self.stored_infos = {}
