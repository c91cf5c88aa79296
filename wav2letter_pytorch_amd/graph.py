"""The training step as a hipGraph.

At N = 32 the host enqueues a step three times faster than the GPU runs it, but the reference's shipped default
(``mid_layers: 1``) and small batches are HOST-bound: ~60-300 launches per step cost more Python time than GPU time.
``GraphedTrainStep`` runs the step eagerly a few times (shapes get tuned, optimizer state and operand packs come into
being), captures ONE step -- zero_grad, forward, CTC, backward, optimizer step -- into a graph through
``torch.cuda.CUDAGraph`` (PyTorch owns streams and the capture; every kernel is this library's) and replays it.

What makes the step capturable:
  * every launch goes to the caller's current stream; the weight-gradient side stream forks from and joins it inside
    backward, so it becomes a parallel branch of the graph;
  * no host synchronisation in the step (the tuners only run while shapes are new, i.e. during the warm-up);
  * dropout: the Philox offset is (unit index) + a step counter that lives in DEVICE memory and is bumped by a captured
    add, so every replay draws fresh masks (w2l_bnact_t.offset_dev);
  * the batch lives in static device buffers; ``__call__`` copies a new batch in (same shapes) and replays.
The learning rate is a kernel argument: re-capture (``recapture()``) after a scheduler step.  Wav2Letter only (Jasper's
masked convolutions take host lengths)."""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, inputs, input_lengths, targets, target_lengths, warmup: int = 3):
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('GraphedTrainStep needs the model on the GPU')
        if type(model).__name__ != 'Wav2Letter':
            raise NotImplementedError('graph capture is implemented for Wav2Letter')
        self.model, self.optimizer = model, optimizer
        self.x = inputs.detach().to(dev, torch.float32).clone()
        self.targets = targets.detach().to(dev, torch.int32).clone()
        self.target_lengths = torch.as_tensor(target_lengths).to(dev, torch.int32).clone()
        self.output_lengths = model.compute_output_lengths(torch.as_tensor(input_lengths)).to(dev, torch.int32)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        model._dropout_counter = self.counter
        self.n_units = len(list(model.conv1ds.children()))
        if hasattr(optimizer, 'overlap'):
            if hasattr(optimizer, 'defer_wgrad'):
                optimizer.defer_wgrad(model, 0)  # a graph holds ONE step: nothing of it may be left for the next forward pass
            if hasattr(optimizer, 'join'):
                optimizer.join()
            optimizer.overlap = False            # inside a graph the updates are a branch of the same step
        self.warmup = warmup
        self.graph = None
        self.loss = None
        self.recapture()

    def _body(self):
        self.optimizer.zero_grad(set_to_none=True)
        out, _ = self.model(self.x, None)
        loss = self.model.criterion(out.transpose(0, 1), self.targets, self.output_lengths, self.target_lengths)
        loss.backward()
        self.optimizer.step()
        self.counter += self.n_units
        return loss.detach()

    def recapture(self):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # eager warm-up off the default stream, as graph capture requires
            for _ in range(self.warmup):
                self._body()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def __call__(self, inputs=None, targets=None, target_lengths=None):
        """replay one step (on a new batch of the captured shapes when given); returns the loss (a static device scalar)"""
        if inputs is not None:
            self.x.copy_(inputs, non_blocking=True)
        if targets is not None:
            self.targets.copy_(targets, non_blocking=True)
        if target_lengths is not None:
            self.target_lengths.copy_(torch.as_tensor(target_lengths), non_blocking=True)
        self.graph.replay()
        return self.loss
