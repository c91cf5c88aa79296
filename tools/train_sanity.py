import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from wav2letter_pytorch_amd import Wav2Letter
from wav2letter_pytorch_amd.defaults import wav2letter_model, synthetic_batch
torch.manual_seed(0)
cfg = wav2letter_model(20)
cfg.optimizer.lr = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-4
if len(sys.argv) > 2:
    cfg.precision = sys.argv[2]          # bf16 | fp32 | fp8 (W2L_FP8_DGRAD / W2L_FP8_WGRAD = 1 force the e4m3 gradients)
model = Wav2Letter(cfg).cuda().train()
opt = model.configure_optimizers()[0][0]
opt.overlap = True
x, il, tg, tl = synthetic_batch(16, 600, seed=3, s_lo=20, s_hi=60)
x = x.cuda(); tg = tg.cuda(); tl = tl.cuda()
ol = model.compute_output_lengths(il).cuda()
t0 = time.time()
for it in range(300):
    opt.zero_grad(set_to_none=True)
    out, _ = model(x, None)
    loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
    loss.backward()
    opt.step()
    if it % 25 == 0 or it == 299:
        print(it, round(float(loss), 4), flush=True)
opt.join()
torch.cuda.synchronize()
print('time', round(time.time() - t0, 1), 'finite', all(torch.isfinite(p).all().item() for p in model.parameters()))
