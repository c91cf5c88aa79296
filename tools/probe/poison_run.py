"""Runs pytest on the given test ids after filling the caching allocator's free blocks with a byte pattern: a kernel that reads
memory it (or its producer) never wrote then sees that pattern instead of the zeros a fresh process gets from the driver.
    python tools/probe/poison_run.py 0xFF tests/test_gpu_model.py::test_x [more ids / pytest args]"""
import sys
import pytest
import torch

pat = int(sys.argv[1], 0)
sizes = [1 << 30] * 6 + [256 << 20] * 8 + [32 << 20] * 16 + [4 << 20] * 32 + [512 << 10] * 64 + [64 << 10] * 128 + [4 << 10] * 256 + [512] * 512
blocks = [torch.full((s,), pat, dtype=torch.uint8, device='cuda') for s in sizes]
torch.cuda.synchronize()
del blocks
sys.exit(pytest.main(sys.argv[2:] + ['-x', '-q', '-p', 'no:cacheprovider']))
