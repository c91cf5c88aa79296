"""Which streams of this process run beside each other?  Prints the w2l_stream_probe matrix over the main stream and N
fresh pool streams (row = stream of the chip-filling kernel, column = stream of the stamp kernel; value = where in the
fill kernel's life the stamp kernel started: ~0 side by side, >= 0.8 queued behind it).  See streams.py / DESIGN.md §7."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    from wav2letter_pytorch_amd import streams as S
    dev = torch.device('cuda', 0)
    torch.zeros(8, device=dev).add_(1)
    ss = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(n)]
    for rep in range(2):
        print(f'pass {rep}: GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES")}')
        print('      ' + ' '.join(f'{j:5d}' for j in range(len(ss))))
        for i, a in enumerate(ss):
            print(f'{i:5d} ' + ' '.join(f'{S.overlap_fraction(a, b, dev):5.2f}' for b in ss))
    for role in ('r1', 'r2', 'r3', 'r4'):
        S.concurrent_stream(dev, role)
    print(S.report)


if __name__ == '__main__':
    main()
