"""GraphedTrainStep: the QAT step of the deform-stage stack as one HIP graph.

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn



class GraphedTrainStep:
    """One training step -- forward, loss, backward, optimizer -- over static input buffers as ONE HIP graph.

    The reference's training loop (quant_main.py -> lib/trains/base_trainer.py:51-80) launches eagerly; the QAT step of
    the three deform stages is 61 kernels of 4-130 us, so eager launches are host-bound (1.4-1.9 ms per step against
    1.05 ms of GPU work).  Captured once, the step replays without the host in the loop.  Every
    sum of the step has a fixed order (section 4.3), so a replayed step is bit-identical to the eager one.

        opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(1.25e-4, device=dev), capturable=True)
        step = GraphedTrainStep(net, opt, loss_fn, example_inputs)                  # loss_fn(net, *inputs) -> scalar
        for batch in loader:
            loss = step(*batch)           # copies the batch into the static buffers, replays; loss: a static tensor
            step.set_lr(schedule(it))     # (a learning rate held as a Python float is a constant of the graph)

    Shapes are fixed at capture; QuantAct running ranges, BatchNorm buffers and the optimizer state advance in place
    exactly as in the eager loop.  `warmup` eager steps run first on a side stream (they DO train: allocator and lazily
    derived tensors settle before capture); the capture records on that same stream.

    SCOPE (round 6).  Validated -- and accepted without `unvalidated=True` -- is the stack of deform stages
    (``pipeline.build_hot_path`` / a quantised ``deconv_layers``): every kernel of that step is this library's, every sum
    has one order, and tests/test_train_step.py::test_graphed_train_step_* show replays bit-identical to the eager step and
    bit-identical from a restored state whatever else the process does in between (a second model built before the capture
    taking its first eager steps, another framework model training, the allocator's free memory filled with NaN).
    A network that also runs PyTorch-ROCm operators under autograd (the whole CoDeNet: backbone and heads) needs
    `unvalidated=True`; what was measured for it (tools/experiments/gts_probe*.py):
      * parameters and gradients of a replay agree with any other replay FROM THE SAME STATE to ~1e-6 of their magnitude --
        with or without other work in between.  That residue is the framework's own backward (atomics), present in two
        eager runs too; after a quantiser amplifies it the loss trajectories of two runs part ways in the fifth digit by
        the second step.  This -- not memory corruption -- is what round 5 recorded as "wrong losses after a second model's
        first steps": poisoning every free byte of the allocator with NaN between replays changes nothing;
      * the scalar LOSS a replay returns can be stale in one situation: loss_fn reduces a large tensor with one
        ``mean()`` / ``sum()`` (ATen's multi-block reduction: a semaphore word zeroed by a memset node), and another model
        takes its FIRST eager step between two replays -- that replay's reduction leaves its output unwritten while the
        gradients and parameters of the same replay are right (the backward of a mean does not read its value).  A
        two-level reduction (``v.square().reshape(-1, 64).sum(1).sum() / v.numel()``: no semaphore) does not show it."""

    def __init__(self, net, optimizer, loss_fn, example_inputs, warmup=3, unvalidated=False):
        if not unvalidated and not self.is_stage_stack(net):
            raise NotImplementedError(
                "GraphedTrainStep is validated (bit-identical replays) for a stack of quantised deform stages only; `net` "
                "holds other modules, whose PyTorch-ROCm kernels under autograd are not run-to-run deterministic. Pass "
                "unvalidated=True to capture it anyway (see the class docstring for what was measured).")
        self.static = [t.detach().clone() for t in example_inputs]
        for t, src in zip(self.static, example_inputs):
            t.requires_grad_(src.requires_grad)
        self._net, self._opt, self._loss_fn = net, optimizer, loss_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # capture ON the warm-up stream: the autograd nodes of the parameters were created there, and a capture stream of
        # its own makes every gradient accumulation a cross-stream branch of the graph (the framework warns about it)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = loss_fn(net, *self.static)
            self.loss.backward()
            optimizer.step()

    @staticmethod
    def is_stage_stack(net):
        """True for a (container of one) Sequential of [quantised deform stage, Sequential(ReLU, QuantAct), Upsample]
        blocks -- what functions/codenet_stage.forward_stage_blocks runs natively and the bit-level tests cover."""
        from ..portable_quantizer.quant_modules import QuantAct, QuantDeformConvWithOffsetScaleBoundPositive
        seq = getattr(net, "deconv_layers", net)
        if not isinstance(seq, nn.Sequential) or len(seq) == 0 or len(seq) % 3:
            return False
        if seq is not net and [m for m in net.children()] != [seq]:
            return False
        mods = list(seq)
        for i in range(0, len(mods), 3):
            q, post, up = mods[i:i + 3]
            if not (isinstance(q, QuantDeformConvWithOffsetScaleBoundPositive) and isinstance(post, nn.Sequential)
                    and len(post) == 2 and isinstance(post[0], nn.ReLU) and isinstance(post[1], QuantAct)
                    and isinstance(up, nn.Upsample)):
                return False
        return True

    def set_lr(self, value, group=None):
        """Write a new learning rate where the captured step reads it: the param group's lr must be a device TENSOR
        (torch.optim with capturable=True accepts one).  A Python float was baked into the graph at capture -- a scheduler
        that assigns ``group['lr'] = float`` changes nothing on replay -- so that case raises instead of silently ignoring."""
        groups = self._opt.param_groups if group is None else [self._opt.param_groups[group]]
        for g in groups:
            if not torch.is_tensor(g["lr"]):
                raise RuntimeError("GraphedTrainStep.set_lr: this optimizer holds its learning rate as a Python float, which "
                                   "the captured graph holds as a constant; construct the optimizer with "
                                   "lr=torch.tensor(value, device=...) (capturable=True) before the capture")
            with torch.no_grad():
                g["lr"].fill_(float(value))

    def eager_step(self):
        """The same step with eager launches (what the capture records), on the static buffers."""
        self._opt.zero_grad(set_to_none=True)
        loss = self._loss_fn(self._net, *self.static)
        loss.backward()
        self._opt.step()
        return loss

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise ValueError("GraphedTrainStep: %d inputs, captured with %d" % (len(inputs), len(self.static)))
        with torch.no_grad():
            for dst, src in zip(self.static, inputs):
                if dst.shape != src.shape:
                    raise ValueError("GraphedTrainStep: input shape %s, captured with %s" % (tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
