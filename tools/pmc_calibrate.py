"""Known byte counts for rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: 256 MiB copied with 16-byte accesses (mode 0) and
with 4-byte accesses (mode 3), five launches each, sources larger than the 256 MiB Infinity Cache in total so that the reads
come from HBM.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; expected per launch: 262 144 KB each way."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << 26                                                   # 256 MiB of floats
bufs = [(torch.empty(n, device=dev).fill_(float(k)), torch.empty(n, device=dev)) for k in range(3)]
lib, stream = _C.lib(), _C.stream_ptr(dev)
for mode in (0, 3):
    for rep in range(5):
        src, dst = bufs[rep % 3]
        _C.check(lib.trl_peak_copy_f32(src.data_ptr(), dst.data_ptr(), n, mode, stream), "copy")
torch.cuda.synchronize()
print("done")
