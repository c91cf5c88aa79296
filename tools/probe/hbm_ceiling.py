import torch, time
def timeit(fn, reps=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for mb in (14, 29, 57, 115, 230, 920):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device='cuda').to(torch.bfloat16)
    b = torch.empty_like(a)
    t_copy = timeit(lambda: b.copy_(a))
    t_sum = timeit(lambda: a.view(torch.int16).sum())          # read-only
    t_fill = timeit(lambda: b.zero_())
    f = a.view(-1, 8)
    print(f'{mb:4d} MB: copy {t_copy:6.1f} us = {2*mb*1.048576/t_copy:5.2f} TB/s (r+w) | sum {t_sum:6.1f} us = {mb*1.048576/t_sum:5.2f} TB/s | fill {t_fill:6.1f} us = {mb*1.048576/t_fill:5.2f} TB/s')
