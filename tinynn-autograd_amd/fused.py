"""Whole-step Dense-MLP trainer: the loop body of examples/mnist/run.py:79-83 as 2L - 2 launches (4 for the 3-layer MNIST net).

Python face of the tnn_mlp_* entry points (csrc/tnn_mlp.cpp).  It owns four flat HBM arenas
(params | grads | m | v, core/optimizer.py:14-15 order) and per-layer activation buffers, and exposes the
parameters as DeviceArray views so a `Net` built from the reference-style layers can share them.

The step can be replayed from a hipGraph (`use_graph=True`): the batch is copied into static staging
buffers and the captured launch sequence — including Adam's device-side bias-correction state — is
replayed with one hipGraphLaunch, which is what makes the dispatch-bound MNIST-size step fast.

Data parallel (comm given): forward + local {max, sum-exp}  ->  all-gather/merge (C2)  ->  loss +
backward with the GLOBAL batch size  ->  in-place all-reduce of grads (+ the loss slot) (C1)  ->  update.
"""

import ctypes

import numpy as np

from . import _lib
from . import device_array as da

_LOSS = {"softmax_nll": 0, "mse": 1}
_OPT = {"sgd": 0, "adam": 1, "momentum": 2, "rmsprop": 3, "adagrad": 4, "adadelta": 5}


class MLPTrainer(object):

    def __init__(self, widths, max_rows, loss="softmax_nll", optimizer="adam", lr=1e-3, beta1=0.9,
                 beta2=0.999, epsilon=1e-8, dtype=np.float32, comm=None, use_graph=False, force_dp=False):
        """optimizer: "sgd", "adam" (beta1, beta2, epsilon) or the reference's other four with their hyper-parameters
        in the same slots — "momentum" (beta1 = momentum), "rmsprop" (beta1 = decay, beta2 = momentum, epsilon),
        "adagrad" (epsilon), "adadelta" (beta1 = decay, epsilon)."""
        self.widths = [int(w) for w in widths]
        self.n_layers = len(self.widths) - 1
        self.max_rows = int(max_rows)
        # Hidden widths that are not multiples of 16 (the reference's own example net: 200-100-70-30) are PADDED to the next
        # multiple inside the arenas so that the 2L - 2 launch step's tiled kernels apply.  Exact, not approximate: a padded
        # unit has zero incoming weights and zero bias (pre-activation 0, activation 0), zero outgoing weights (its dz is a
        # sum of zeros), hence zero gradients everywhere and a zero Adam step — it stays zero for ever.  `param_view` /
        # `grad_view` / `get_parameters` present the logical shapes; `params` / `state_dict` are the padded arenas.
        self._pwidths = list(self.widths)
        if (loss == "softmax_nll" and optimizer == "adam" and self.n_layers >= 3
                and not (isinstance(dtype, str) and dtype in ("bfloat16", "bf16")) and np.dtype(dtype) == np.float32
                and any(w % 16 for w in self.widths[1:-1]) and self.widths[-1] <= 16
                and (self.widths[-2] + 15) // 16 * 16 <= 256):
            self._pwidths = [self.widths[0]] + [(w + 15) // 16 * 16 for w in self.widths[1:-1]] + [self.widths[-1]]
        self.padded = self._pwidths != self.widths
        # dtype "bfloat16": bf16 inputs / activations / working weights, fp32 master weights + gradients + Adam
        # state (BASELINE.json configs[4]).  The arenas exposed below are then the fp32 ones.
        self.bf16 = isinstance(dtype, str) and dtype in ("bfloat16", "bf16")
        self.dtype = np.dtype(np.float32) if self.bf16 else np.dtype(dtype)
        # force_dp keeps the sharded code path (stats exchange, arena all-reduce) even at world size 1, which
        # is how the RCCL path is exercised on a single-GPU box
        self.comm = comm if (comm is not None and (comm.world > 1 or force_dp)) else None
        self.use_graph = bool(use_graph) and self.comm is None
        self._lib = _lib.get()
        self._h = ctypes.c_void_p()
        code = _lib.BF16 if self.bf16 else da._CODE[self.dtype]
        self._lib.mlp_create(self.n_layers, da._i64arr(self._pwidths), self.max_rows, _LOSS[loss],
                             _OPT[optimizer], float(lr), float(beta1), float(beta2), float(epsilon),
                             code, ctypes.byref(self._h))
        p, g, m, v = (ctypes.c_void_p() for _ in range(4))
        n = ctypes.c_int64(0)
        self._lib.mlp_arena(self._h, ctypes.byref(p), ctypes.byref(g), ctypes.byref(m), ctypes.byref(v),
                            ctypes.byref(n))
        # The four arenas AS STORED (padded widths): what the kernels, the collectives and a checkpoint work on.  `params` /
        # `grads` / `adam_m` / `adam_v` / `n_params` below are the LOGICAL view (the reference optimizer's flat order and
        # count, core/optimizer.py:14-15): the arenas themselves unless hidden widths are padded, read-only copies then.
        self.arena_size = n.value
        self.arena_params = da.from_ptr(p.value, (self.arena_size,), self.dtype, self)
        self.arena_grads = da.from_ptr(g.value, (self.arena_size,), self.dtype, self)
        self._grads_and_loss = da.from_ptr(g.value, (self.arena_size + 1,), self.dtype, self)
        self.loss_slot = da.from_ptr(g.value + self.arena_size * self.dtype.itemsize, (), self.dtype, self)
        self.arena_m = da.from_ptr(m.value, (self.arena_size,), self.dtype, self)
        self.arena_v = da.from_ptr(v.value, (self.arena_size,), self.dtype, self)
        self.n_params = sum(self.widths[l] * self.widths[l + 1] + self.widths[l + 1] for l in range(self.n_layers))
        self._graph = None
        self._graph_rows = None
        self._x_stage = None
        self._y_stage = None
        self._stats = da.empty((2,), self.dtype)

    @property
    def params(self):
        return self.flat_parameters()

    @property
    def grads(self):
        return self.flat_parameters(self.arena_grads)

    @property
    def adam_m(self):
        return self.flat_parameters(self.arena_m)

    @property
    def adam_v(self):
        return self.flat_parameters(self.arena_v)

    def keep_grads(self, keep=True):
        """keep=False: `step` may consume weight gradients where they are produced (bf16 trainer: Adam in the epilogue of
        the dW GEMM) without writing them to the gradient arena — `grad_view(l, "w")` is then undefined after `step`
        (bias gradients and the loss are still written).  Default: every gradient is stored, as in the reference
        (core/model.py:24-33 exposes them after backward)."""
        self._lib.mlp_keep_grads(self._h, 1 if keep else 0)
        return self

    def __del__(self):
        try:
            self._graph = None
            if _lib._lib is not None and self._h:
                _lib._lib.mlp_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _offset(self, layer, which):
        off, cnt = ctypes.c_int64(0), ctypes.c_int64(0)
        self._lib.mlp_param_offset(self._h, layer, which, ctypes.byref(off), ctypes.byref(cnt))
        return off.value, cnt.value

    def masters_sharded(self):
        """World size the fp32 master / moment arenas are currently sharded over (bf16 data-parallel trainer after
        sharded-optimizer steps), 0 when they are whole."""
        w = ctypes.c_int(0)
        self._lib.mlp_masters_sharded(self._h, ctypes.byref(w))
        return w.value

    def gather_masters(self):
        """COLLECTIVE (every rank calls it): make the fp32 parameter / Adam arenas whole on every rank after
        sharded-optimizer steps.  `state_dict`, `save` and `get_parameters` refuse to run on sharded masters unless told
        `collective=True` (then they call this themselves — on EVERY rank, or the ranks that did call hang)."""
        self._lib.mlp_gather_masters(self._h)
        return self

    def _whole_masters(self, collective, what):
        if self.masters_sharded():
            if not collective:
                raise RuntimeError("%s: the fp32 masters are sharded over %d ranks (sharded-optimizer steps) — making them "
                                   "whole is a COLLECTIVE.  Call it with collective=True on every rank, or call "
                                   "gather_masters() on every rank first (an `if rank == 0: trainer.save(...)` would hang in "
                                   "the all-gather)" % (what, self.masters_sharded()))
            self.gather_masters()

    def param_view(self, layer, key, arena=None):
        if key == "w" and arena is not self.arena_grads and self.masters_sharded():      # master weights AND both moments
            raise RuntimeError("the fp32 master weights / Adam moments are sharded over %d ranks (sharded-optimizer steps): "
                               "call gather_masters() on every rank first, or read weights_bf16()" % self.masters_sharded())
        view = self._view(layer, key, arena)
        if self.padded and view.shape != self._shape(layer, key, self.widths):
            rows, cols = self._shape(layer, key, self.widths)
            view = view[:rows, :cols]                  # a COPY of the logical block (the padding is zeros) ...
            view._tag = da.READONLY_COPY               # ... that refuses assignment: write through set_param
        return view

    def set_param(self, layer, key, value, arena=None):
        """Write one parameter block ("w" [in, out] or "b" [1, out], logical shape; host or device values) — the way to assign
        on a padded trainer, where `param_view` hands out read-only copies (an unpadded trainer's views can also be assigned
        in place).  bf16 trainers: call sync() afterwards (or use set_parameters)."""
        view = self._view(layer, key, arena)
        rows, cols = self._shape(layer, key, self.widths)
        src = value.values if hasattr(value, "values") and not isinstance(value, np.ndarray) else value
        if view.shape != (rows, cols):                 # padded block: zeros around the logical values
            host = np.zeros(view.shape, self.dtype)
            host[:rows, :cols] = np.asarray(src, dtype=self.dtype).reshape(rows, cols)
            view[...] = da.asarray(host)
        else:
            view[...] = da.asarray(src, dtype=self.dtype).reshape(view.shape)

    @staticmethod
    def _shape(layer, key, widths):
        return (widths[layer], widths[layer + 1]) if key == "w" else (1, widths[layer + 1])

    def _view(self, layer, key, arena=None):
        """The parameter's block of the arena as stored (padded widths)."""
        off, cnt = self._offset(layer, 0 if key == "w" else 1)
        base = self.arena_params if arena is None else arena
        return base[off:off + cnt].reshape(self._shape(layer, key, self._pwidths))

    def flat_parameters(self, arena=None):
        """The parameters (or another arena) as ONE flat vector in the reference optimizer's order (layer by layer, "w" then
        "b": core/optimizer.py:14-15) with the logical shapes — the arena itself unless hidden widths are padded."""
        if not self.padded:
            return self.arena_params if arena is None else arena
        flat = da.asarray(np.concatenate([np.asarray(self.param_view(l, k, arena)).ravel()
                                          for l in range(self.n_layers) for k in ("w", "b")]))
        flat._tag = da.READONLY_COPY
        return flat

    def grad_view(self, layer, key):
        return self.param_view(layer, key, arena=self.arena_grads)

    def set_parameters(self, layers):
        """layers: list of {"w": array [in,out], "b": array [1,out]} (host or device)."""
        for i, layer in enumerate(layers):
            for key in ("w", "b"):
                self.set_param(i, key, layer[key])    # the whole arena is rewritten: no need for it to be whole before
        self._lib.mlp_sync_params(self._h)           # bf16 mode: refresh the working copies W, W^T

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self, collective=False):
        """Everything a resumed run needs, as host arrays: parameters, the optimizer's two state arenas and Adam's
        beta powers (the reference's Model.save only pickles the parameters and its load is broken).  After
        sharded-optimizer steps (bf16 data-parallel trainer) the masters must be made whole first — a COLLECTIVE: pass
        collective=True on EVERY rank (or call gather_masters() on every rank beforehand); without it this raises instead of
        hanging the ranks that did call."""
        self._whole_masters(collective, "state_dict")
        pows = ctypes.c_void_p()
        self._lib.mlp_optimizer_state(self._h, ctypes.byref(pows))
        return {"widths": list(self.widths), "padded_widths": list(self._pwidths),
                "dtype": "bfloat16" if self.bf16 else self.dtype.name,
                "params": np.asarray(self.arena_params).copy(), "m": np.asarray(self.arena_m).copy(),
                "v": np.asarray(self.arena_v).copy(),
                "pows": np.asarray(da.from_ptr(pows.value, (4,), np.float64, self)).copy()}

    def load_state_dict(self, state):
        if list(state["widths"]) != list(self.widths):
            raise ValueError("checkpoint is for widths %s, this trainer has %s" % (state["widths"], self.widths))
        if list(state.get("padded_widths", state["widths"])) != list(self._pwidths):
            raise ValueError("checkpoint arenas are laid out for widths %s, this trainer's for %s"
                             % (list(state.get("padded_widths", state["widths"])), self._pwidths))
        pows = ctypes.c_void_p()
        self._lib.mlp_optimizer_state(self._h, ctypes.byref(pows))
        self.arena_params[...] = da.asarray(np.asarray(state["params"]), dtype=self.dtype)
        self.arena_m[...] = da.asarray(np.asarray(state["m"]), dtype=self.dtype)
        self.arena_v[...] = da.asarray(np.asarray(state["v"]), dtype=self.dtype)
        da.from_ptr(pows.value, (4,), np.float64, self)[...] = da.asarray(np.asarray(state["pows"]), dtype=np.float64)
        self._lib.mlp_sync_params(self._h)           # bf16 mode: refresh the working copies
        self._graph = None                            # a captured single-step graph holds no state, but be safe

    def save(self, path, collective=False):
        np.savez(path, **{k: np.asarray(v) for k, v in self.state_dict(collective=collective).items()})

    def load(self, path):
        with np.load(path, allow_pickle=False) as f:
            self.load_state_dict({k: (f[k].tolist() if k in ("widths", "padded_widths") else (str(f[k]) if k == "dtype" else f[k]))
                                  for k in f.files})

    def get_parameters(self, collective=False):
        self._whole_masters(collective, "get_parameters")
        return [{"w": self.param_view(i, "w"), "b": self.param_view(i, "b")} for i in range(self.n_layers)]

    def weights_bf16(self, layer=None):
        """bf16 trainer: the bf16 working copy of the parameters as raw uint16 bit patterns — the whole arena, or layer
        `layer`'s [in, out] weight matrix.  In the data-parallel sharded-optimizer step (tnn_mlp_step_sharded on a bf16
        trainer) this is the complete, rank-identical copy; each rank's fp32 master arena (`params`, `adam_m`, `adam_v`) is
        then authoritative for its own row slice of every weight matrix only."""
        p = ctypes.c_void_p()
        self._lib.mlp_bf16_weights(self._h, ctypes.byref(p))
        arena = da.from_ptr(p.value, (self.arena_size,), np.uint16, self)
        if layer is None:
            return arena
        off, cnt = self._offset(layer, 0)
        return arena[off:off + cnt].reshape((self.widths[layer], self.widths[layer + 1]))

    def activation(self, layer, rows):
        p = ctypes.c_void_p()
        self._lib.mlp_activation(self._h, layer, ctypes.byref(p))
        act = da.from_ptr(p.value, (rows, self._pwidths[layer + 1]), np.uint16 if self.bf16 else self.dtype, self)
        return act[:, :self.widths[layer + 1]] if self._pwidths[layer + 1] != self.widths[layer + 1] else act

    # ------------------------------------------------------------------ compute
    def _prep16(self, x, y=None):
        from . import bf16 as _bf16

        def as16(a):
            a = da.asarray(a)
            return a._contig() if a.dtype == np.uint16 else _bf16.to_bf16(a)     # pre-cast data is used as is
        x = as16(x)
        rows = x.shape[0]
        if x.ndim != 2 or x.shape[1] != self.widths[0] or not 0 < rows <= self.max_rows:
            raise ValueError("inputs %s do not fit a [<=%d, %d] batch" % (x.shape, self.max_rows, self.widths[0]))
        if rows % 64:
            raise ValueError("the bf16 path needs a batch that is a multiple of 64 rows (K of the dW GEMM)")
        if y is not None:
            y = as16(y)
            if y.shape != (rows, self.widths[-1]):
                raise ValueError("targets %s must be [%d, %d]" % (y.shape, rows, self.widths[-1]))
        return x, y, rows

    def _prep(self, x, y=None):
        if self.bf16:
            return self._prep16(x, y)
        x = da.asarray(x, dtype=self.dtype)._contig()
        rows = x.shape[0]
        if x.ndim != 2 or x.shape[1] != self.widths[0] or not 0 < rows <= self.max_rows:
            raise ValueError("inputs %s do not fit a [<=%d, %d] batch" % (x.shape, self.max_rows, self.widths[0]))
        if y is not None:
            y = da.asarray(y, dtype=self.dtype)._contig()
            if y.shape != (rows, self.widths[-1]):
                raise ValueError("targets %s must be [%d, %d]" % (y.shape, rows, self.widths[-1]))
        return x, y, rows

    def forward(self, x):
        x, _, rows = self._prep(x)
        if self.bf16:
            from . import bf16 as _bf16
            out16 = da.empty((rows, self.widths[-1]), np.uint16)
            self._lib.mlp_forward(self._h, x._ptr, rows, out16._ptr)
            return _bf16.to_f32(out16)
        out = da.empty((rows, self.widths[-1]), self.dtype)
        self._lib.mlp_forward(self._h, x._ptr, rows, out._ptr)
        return out

    def step(self, x, y):
        """One training step; returns the loss as a 0-d DeviceArray (no host sync)."""
        x, y, rows = self._prep(x, y)
        if self.comm is not None:
            return self._step_dp(x, y, rows)
        if self.use_graph:
            return self._step_graph(x, y, rows)
        self._lib.mlp_step(self._h, x._ptr, y._ptr, rows, None)
        return self.loss_slot

    def capture_steps(self, batches):
        """Capture one training step per (x, y) batch into a single replayable hipGraph."""
        return StepGraph(self, batches)

    def _step_dp(self, x, y, rows):
        lib = self._lib
        from .dist import DeviceCommunicator
        if isinstance(self.comm, DeviceCommunicator):
            lib.mlp_step_sharded(self._h, x._ptr, y._ptr, rows, None)      # all phases + both collectives in C
            return self.loss_slot
        lib.mlp_forward_stats(self._h, x._ptr, rows, self._stats._ptr)
        stats = self.comm.merge_softmax_stats(self._stats)
        lib.mlp_backward(self._h, x._ptr, y._ptr, rows, rows * self.comm.world, stats._ptr, None)
        self.comm.allreduce(self._grads_and_loss)      # gradients and the loss share one collective
        lib.mlp_update(self._h)
        return self.loss_slot

    def _step_graph(self, x, y, rows):
        lib = self._lib
        if self._graph is None or self._graph_rows != rows:
            self._x_stage = da.empty((rows, self.widths[0]), self.dtype)
            self._y_stage = da.empty((rows, self.widths[-1]), self.dtype)
            self._x_stage[...] = x
            self._y_stage[...] = y
            # one eager step first would advance the optimizer; capture records without executing
            g = _lib.Graph()
            with g:
                lib.mlp_step(self._h, self._x_stage._ptr, self._y_stage._ptr, rows, None)
            self._graph, self._graph_rows = g, rows
        else:
            lib.memcpy_d2d(self._x_stage._ptr, x._ptr, x.nbytes)
            lib.memcpy_d2d(self._y_stage._ptr, y._ptr, y.nbytes)
        self._graph.launch()
        return self.loss_slot


class StepGraph(object):
    """hipGraph of several consecutive training steps, each bound to its own HBM-resident batch.

    A hipGraph bakes kernel arguments in, so replaying ONE captured step on a new batch needs the batch
    copied into staging buffers first.  When the batches of an epoch already sit in HBM at fixed addresses
    (row slices of the resident dataset, utils/data_iterator.py:30-33) a whole run of steps can be captured
    once and replayed with a single hipGraphLaunch: no staging copies and no per-step host work.  Adam's
    bias-correction state lives on the device, so every replay continues the optimizer correctly.  The loss of
    step i lands in `losses[i]` (the loss_list of examples/mnist/run.py:84, kept in HBM)."""

    def __init__(self, trainer, batches):
        sharded = trainer.comm is not None
        if sharded:
            from .dist import DeviceCommunicator
            if not isinstance(trainer.comm, DeviceCommunicator):
                raise ValueError("only device-side collectives (RCCL / xGMI peer-to-peer) can be captured into a hipGraph")
        self.trainer = trainer
        self.batches = [trainer._prep(x, y) for x, y in batches]      # keeps the buffers alive
        self.losses = da.empty((len(self.batches),), trainer.dtype)
        lib, esz = trainer._lib, trainer.dtype.itemsize
        self._graph = _lib.Graph()
        with self._graph:
            for i, (x, y, rows) in enumerate(self.batches):
                if sharded:   # forward | all-gather + merge | backward | all-reduce | update, collectives captured too
                    lib.mlp_step_sharded(trainer._h, x._ptr, y._ptr, rows, self.losses._ptr + i * esz)
                else:
                    lib.mlp_step(trainer._h, x._ptr, y._ptr, rows, self.losses._ptr + i * esz)

    def __len__(self):
        return len(self.batches)

    def launch(self):
        """Run all captured steps (asynchronous); returns the device array of their losses.  A data-parallel graph
        first looks at the peer-to-peer transport's failure word (host-pinned, no sync): captured collectives of a dead
        transport would only be discarded on the device, so the replay is refused loudly instead."""
        comm = self.trainer.comm
        if comm is not None and hasattr(comm, "p2p_failed") and comm.p2p_failed():
            from .dist import PeerTimeout
            raise PeerTimeout("refusing to replay a data-parallel step graph: an xGMI peer-to-peer barrier timed out "
                              "earlier (steps since then were discarded); call comm.check() on every rank")
        self._graph.launch()
        return self.losses


def trainer_from_net(net, max_rows, loss="softmax_nll", optimizer="adam", lr=1e-3, **kwargs):
    """Build an MLPTrainer from an initialised Dense/ReLU `Net` (parameters are copied into the arena)."""
    from .core.layers import Dense, ReLU
    dense = [l for l in net.layers if isinstance(l, Dense)]
    others = [l for l in net.layers if not isinstance(l, Dense)]
    if not dense or any(not isinstance(l, ReLU) for l in others) or len(others) != len(dense) - 1:
        raise ValueError("trainer_from_net supports Dense layers separated by ReLU only")
    if any(not l.is_init for l in dense):
        raise ValueError("initialise the net first (pass num_in to Dense or run one forward)")
    widths = [dense[0].params["w"].shape[0]] + [l.params["w"].shape[1] for l in dense]
    trainer = MLPTrainer(widths, max_rows, loss=loss, optimizer=optimizer, lr=lr, **kwargs)
    trainer.set_parameters([l.params for l in dense])
    return trainer
