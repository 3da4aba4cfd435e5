"""Which Box-Muller contraction does PyTorch's build of rocRAND use?  Fills through aesmc_philox_normal_fill with
both variants and compares with torch.empty(n).normal_() bit for bit (prints a table; no assertion)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import aesmc_amd  # noqa: F401
from aesmc_amd import _kernels, _lib, _philox

device = torch.device("cuda", 0)
lib = _kernels.get()._lib
props = torch.cuda.get_device_properties(0)
print("CUs", props.multi_processor_count, "maxThreadsPerCU", props.max_threads_per_multi_processor)
gen = torch.cuda.default_generators[0]
for seed, warm in ((0, 0), (1234, 3), (2 ** 40 + 17, 1)):
    torch.manual_seed(seed)
    for _ in range(warm):
        torch.randn(1000, device=device)
    for numel in (1, 7, 255, 256, 1000, 2 ** 19 - 1, 2 ** 19 + 5, 5 * 2 ** 19 + 123, 256 * 1024 * 10, 1024 * 4096 * 10):
        state = gen.get_state()
        offset = gen.get_offset()
        want = torch.empty(numel, device=device).normal_()
        after = gen.get_offset()
        threads = _philox.launch_threads(numel, device)
        predicted = offset + _philox.consumed(numel, threads)
        row = [seed, numel, threads, offset, after, predicted == after]
        for variant in (0, 1):
            got = torch.empty(numel, device=device)
            status = lib.aesmc_philox_normal_fill(got.data_ptr(), numel, gen.initial_seed(), offset, threads, variant, None,
                                                  torch.cuda.current_stream().cuda_stream)
            assert status == 0, status
            same = (got.view(torch.int32) == want.view(torch.int32))
            row.append(int((~same).sum().item()))
            if not bool(same.all()):
                bad = (~same).nonzero().flatten()[:3].tolist()
                row.append([(i, float(got[i]), float(want[i])) for i in bad])
        print(row, flush=True)
