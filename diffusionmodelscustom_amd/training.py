"""The body of the reference's training loop as one replayable hipGraph.

``TrainingPipeline_general.train`` (reference training_DANRA_conditional.py:509-628) runs, per batch:
``model.zero_grad()`` -> ``t = sampleTimesteps(B)`` -> ``x_t, noise = noiseImage(images, t)`` ->
``predicted = model(x_t, t, seasons, cond_images, lsm, topo)`` -> ``loss = lossfunc(predicted, noise)`` ->
``loss.backward()`` -> ``optimizer.step()``.  On a 64x64 batch that is about a thousand small kernels whose enqueueing by
the host, not their execution, bounds the step.  ``GraphedTrainStep`` captures exactly that sequence once (through
``torch.cuda.CUDAGraph``, i.e. a hipGraph) and replays it per batch: same kernels, same arithmetic, no host work between them.

The reference's loop itself is untouched by this module: a training script opts in by calling the object instead of
writing the seven statements out.
"""
from typing import Dict, Optional

import torch

__all__ = ["GraphedTrainStep", "eager_train_step"]

_PLANES = ("seasons", "cond_images", "lsm", "topo")


def eager_train_step(model, optimizer, diffusion_utils, lossfunc, images, seasons=None, cond_images=None, lsm=None,
                     topo=None, t=None, noise=None):
    """The seven statements of the reference's loop body (training_DANRA_conditional.py:608-624), with the timestep
    draw and the Gaussian draw replaceable by given tensors (for reproducible comparisons).  Returns the loss tensor."""
    model.zero_grad()
    if t is None:
        t = diffusion_utils.sampleTimesteps(images.shape[0])
    x_t, noise = diffusion_utils.noiseImage(images, t, noise) if noise is not None else diffusion_utils.noiseImage(images, t)
    predicted = model(x_t, t, seasons, cond_images, lsm, topo)
    loss = lossfunc(predicted, noise)
    loss.backward()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """``step(images, seasons=..., cond_images=..., lsm=..., topo=...) -> loss`` replaying one captured training step.

    * ``example`` fixes the shapes: a dict with ``images`` and whichever of ``seasons / cond_images / lsm / topo`` the
      model takes (CUDA tensors).  Later calls must pass tensors of the same shapes (a last, smaller batch of an epoch
      goes through ``eager_train_step``).
    * An optimizer that has a ``capturable`` switch (Adam, AdamW, ...) must be built with ``capturable=True``; add
      ``fused=True`` (one multi-tensor kernel per step: the unfused capturable AdamW launches ~300 scalar copy kernels,
      1 ms of a 9 ms step).  The learning rate is baked into the graph unless it is a tensor.
    * Timesteps and noise are drawn on the device inside the graph (``torch.randint`` / ``torch.randn_like`` under
      torch's graph-safe generator), or - ``external_draws=True`` - taken from the ``t=`` / ``noise=`` arguments of
      every call.
    * Warm-up (required before capture, so that every lazy initialisation and solver search has happened) runs
      ``warmup`` real steps on the example batch and then puts parameters, buffers, the optimizer's state (moments and
      step counters restored by ``load_state_dict`` included) and the random generators' states back, so that
      constructing the object does not train the model.  SGD with momentum AND dampening must have taken a step before.
    * The returned loss is a device tensor that the next call overwrites; ``.item()`` it (one host sync, as in the
      reference's loop) or accumulate it on the device.
    """

    def __init__(self, model, optimizer, diffusion_utils, lossfunc, example: Dict[str, torch.Tensor], warmup: int = 3,
                 external_draws: bool = False):
        images = example["images"]
        if not images.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA tensors")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise RuntimeError("the optimizer must be built with capturable=True to be replayed from a graph")
        self.model, self.optimizer, self.du, self.lossfunc = model, optimizer, diffusion_utils, lossfunc
        self.external_draws = bool(external_draws)
        self.static = {k: v.detach().clone() for k, v in example.items() if v is not None}
        unknown = set(self.static) - {"images", "t", "noise", *_PLANES}
        if unknown:
            raise KeyError(f"unknown batch entries {sorted(unknown)}")
        B = images.shape[0]
        if self.external_draws:
            self.static.setdefault("t", torch.ones(B, dtype=torch.long, device=images.device))
            self.static.setdefault("noise", torch.zeros_like(images))

        saved_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # what the optimizer knew before the warm-up (e.g. moments restored by load_state_dict), per parameter
        saved_opt = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                     for p, st in optimizer.state.items()}
        if not saved_opt:
            for group in optimizer.param_groups:
                if group.get("momentum", 0) and group.get("dampening", 0):
                    # SGD's first step copies the gradient into a momentum buffer it creates; a buffer that exists and
                    # holds zeros gets (1 - dampening) * gradient instead, so the warm-up cannot be undone by zero-filling
                    raise RuntimeError("SGD with momentum and dampening != 0 must have taken one step before it is "
                                       "captured (its first step differs from the later ones)")
        cpu_rng = torch.get_rng_state()
        dev_rng = torch.cuda.get_rng_state(images.device)
        side = torch.cuda.Stream(device=images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._body()
        torch.cuda.current_stream(images.device).wait_stream(side)
        torch.cuda.synchronize(images.device)
        # undo the warm-up, in place (the graph will refer to these very tensors): parameters and buffers, the optimizer
        # state (what it held before; moments and step counters the warm-up created go back to zero, which is where a
        # fresh optimizer starts); the generators' states follow after the capture
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(saved_model[k])
            for p, state in optimizer.state.items():
                before = saved_opt.get(p)
                for k, v in list(state.items()):
                    if before is not None and k in before:
                        if torch.is_tensor(v):
                            v.copy_(before[k])
                        else:
                            state[k] = before[k]
                    elif torch.is_tensor(v):
                        v.zero_()
        model.zero_grad(set_to_none=True)

        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        # warm-up and capture drew from the generators (the capture only records its draws, it does not make them)
        torch.set_rng_state(cpu_rng)
        torch.cuda.set_rng_state(dev_rng, images.device)

    def _body(self):
        s = self.static
        model = self.model
        # model.zero_grad() as "forget the gradients": backward then writes each gradient where autograd produced it
        # (inside the graph's own memory pool, so at the same address in every replay) instead of zero-filling ~180
        # tensors and adding into them
        for p in model.parameters():
            p.grad = None
        if self.external_draws:
            t, noise = s["t"], s["noise"]
            x_t, noise = self.du.noiseImage(s["images"], t, noise)
        else:
            B = s["images"].shape[0]
            t = torch.randint(1, self.du.n_timesteps, (B,), device=s["images"].device)  # sampleTimesteps, on the device
            x_t, noise = self.du.noiseImage(s["images"], t, torch.randn_like(s["images"]))
        predicted = model(x_t, t, s.get("seasons"), s.get("cond_images"), s.get("lsm"), s.get("topo"))
        loss = self.lossfunc(predicted, noise)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, images, seasons=None, cond_images=None, lsm=None, topo=None, t: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None):
        given = dict(images=images, seasons=seasons, cond_images=cond_images, lsm=lsm, topo=topo, t=t, noise=noise)
        for k, dst in self.static.items():
            src = given.get(k)
            if src is None:
                if k in ("t", "noise") and not self.external_draws:
                    continue
                raise RuntimeError(f"batch entry '{k}' was part of the captured step and must be passed")
            if src.shape != dst.shape:
                raise RuntimeError(f"'{k}' has shape {tuple(src.shape)}, the captured step was built for {tuple(dst.shape)}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replay rewrites parameters and buffers in place without touching their version counters, which is what
        # DiffusionNet watches to know when its packed inference weights are stale: say so explicitly
        if hasattr(self.model, "_hip_signature"):
            self.model._hip_signature = None
        return self.loss
