"""Worker of test_gpu_model.py::test_lightning_base_class_branch: a MINIMAL stand-in for pytorch_lightning (a LightningModule
that is an nn.Module with log_dict / optimizers, a Trainer.fit doing automatic optimisation in Lightning's hook order) is put
into sys.modules BEFORE this package is imported, so that base_asr_models takes its `_Base = ptl.LightningModule` branch
(base_asr_models.py:15-17 of the reference: the class IS a LightningModule there) and the model is driven the way
/root/reference/train.py:34-37 drives it: Trainer(**cfg.trainer).fit(model, train_loader, val_loader).  Nothing here is the
reference's or Lightning's code; the image has no Lightning to test against."""
import json
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = None
        self.log_calls = []

    def log_dict(self, d, *args, **kwargs):
        self.log_calls.append(dict(d))            # tensors are taken as they are (Lightning detaches, never syncs here)

    def optimizers(self):
        return self.trainer.optimizers[0]

    # hooks a LightningModule has; the loop below calls them where Lightning does
    def on_train_batch_end(self, outputs, batch, batch_idx):
        pass

    def on_train_epoch_end(self):
        pass


class Trainer:
    def __init__(self, max_epochs=1, **unused):
        self.max_epochs = max_epochs
        self.optimizers, self.schedulers = [], []
        self.grads_seen = []

    def fit(self, model, train_dataloader, val_dataloader=None):
        model.trainer = self
        model = model.cuda()
        self.optimizers, self.schedulers = model.configure_optimizers()
        opt = self.optimizers[0]
        for _ in range(self.max_epochs):
            model.train()
            for i, batch in enumerate(train_dataloader):
                opt.zero_grad()
                loss = model.training_step(batch, i)
                loss.backward()
                # automatic optimisation reads p.grad of EVERY parameter (gradient clipping, norm logging): none may be held back
                self.grads_seen.append(all(p.grad is not None for p in model.parameters()))
                opt.step()
                model.on_train_batch_end(loss, batch, i)
            model.on_train_epoch_end()
            for s in self.schedulers:
                s.step()
            if val_dataloader is not None:
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val_dataloader):
                        model.validation_step(batch, i)
        return model


def main():
    ptl = types.ModuleType('pytorch_lightning')
    ptl.LightningModule, ptl.Trainer = LightningModule, Trainer
    sys.modules['pytorch_lightning'] = ptl
    sync = len(sys.argv) > 1 and sys.argv[1] == 'sync'
    if sync:
        os.environ['W2L_SYNC_METRICS'] = '1'
    from gpu_helpers import build_w2l
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import base_asr_models as B
    assert B._Base is LightningModule and issubclass(B.ConvCTCASR, LightningModule)
    layers = [(128, 11, 2, 1, 0.0), (128, 11, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=21)
    model = build_w2l(layers, sd, 'bf16')
    model._cfg.optimizer.lr = 0.05
    x, il, tg, tl = O.synthetic_batch(4, 200, seed=22, s_lo=5, s_hi=12)
    texts = tuple(''.join(O.ENGLISH_LOWERCASE[int(i)] for i in tg[n, :int(tl[n])]) for n in range(4))
    batch = (x, il, tg, tl, ('a', 'b', 'c', 'd'), texts)
    p0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    import pytorch_lightning
    tr = pytorch_lightning.Trainer(max_epochs=2, default_root_dir='.', gpus=1)
    tr.fit(model, [batch] * 5, [batch])
    torch.cuda.synchronize()
    train = [c for c in model.log_calls if 'train_loss' in c]
    val = [c for c in model.log_calls if 'val_loss' in c]
    tensor_valued = any(torch.is_tensor(v) for c in model.log_calls for v in c.values())
    moved = sum(int(not torch.equal(v.cpu(), p0[k].cpu())) for k, v in model.state_dict().items() if v.dtype.is_floating_point)
    print(json.dumps({'train_loss': [float(c['train_loss']) for c in train], 'n_val': len(val),
                      'train_keys': sorted(train[-1]), 'val_keys': sorted(val[-1]), 'tensor_valued': tensor_valued,
                      'grads_seen': tr.grads_seen, 'moved': moved, 'lr': [float(c['learning_rate']) for c in train],
                      'optimizer': type(tr.optimizers[0]).__name__, 'pending': len(getattr(model, '_pending_metrics', []))}))


if __name__ == '__main__':
    main()
