"""The subset of the Keras surface that the reference's mains touch, backed by the HIP engine (GaitCore).

Nothing here computes on the hot path: these classes carry configuration, drive the training loop on the host and
move numpy batches to HBM.  Surfaces mirrored:
  * `optimizers.Adam(lr=...)`                          mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:227
  * `Maximum`, `Average`, `sign_max(**kwargs)`          nets/mj_uwyhNets_ba.py:814,1189; mains/...CasiaB.py:169-178
  * `Model.fit / predict / save / save_weights / load_weights / get_layer / summary / optimizer.lr / loss / input`
    as used by nets/mj_uwyhNets_ba.py:937-999 and the mains (grep counts in SURVEY.md section 8b)
  * `History.epoch`, `History.history[...]`             mains/mj_trainUWYHGaitNet_DataGen_1mod.py:544,551,614-637
"""
from __future__ import annotations

import json
import os

import numpy as np


# ---------------------------------------------------------------------------------------------------------
# optimizers
# ---------------------------------------------------------------------------------------------------------
class _Optimizer:
    def get_config(self):
        return dict(self.__dict__, name=type(self).__name__)


class Adam(_Optimizer):
    """keras.optimizers.Adam defaults (epsilon 1e-7).  The update itself is the ugn_adam_step HIP kernel."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, learning_rate=None, **kwargs):
        if amsgrad:
            raise NotImplementedError("AMSGrad is not part of the MI355X hot path (SURVEY.md section 8: out of scope)")
        self.lr = float(lr if learning_rate is None else learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)

    @property
    def learning_rate(self):
        return self.lr

    @learning_rate.setter
    def learning_rate(self, v):
        self.lr = float(v)


class SGD(_Optimizer):
    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False, learning_rate=None, **kwargs):
        self.lr = float(lr if learning_rate is None else learning_rate)
        self.momentum, self.decay, self.nesterov = float(momentum), float(decay), bool(nesterov)


class optimizers:  # namespace, so that `optimizers.Adam(lr=lr)` reads like the reference
    Adam = Adam
    SGD = SGD


# ---------------------------------------------------------------------------------------------------------
# fMerge factories
# ---------------------------------------------------------------------------------------------------------
class _Fusion:
    mode = None

    def __init__(self, name="fusion", **kwargs):
        self.name = name

    def __call__(self, tensors):
        raise TypeError("fusion layers are descriptors here: the merge runs inside ugn_gate_fuse_fwd on the GPU")


class Maximum(_Fusion):
    """keras.layers.Maximum stand-in (fMerge default, nets/mj_uwyhNets_ba.py:585)."""
    mode = "max"


class Average(_Fusion):
    """keras.layers.Average stand-in."""
    mode = "avg"


class _SignMax(_Fusion):
    mode = "sign_max"


def sign_max(**kwargs):
    """Factory with the reference's signature (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:169-178):
    elementwise pick of the modality with the largest |value|, sign kept, first index on ties."""
    return _SignMax(**kwargs)


def fusion_mode(fmerge):
    """Accepts the class/factory (as `eval(mergefun)` yields in the reference) or an instance."""
    obj = fmerge(name="fusion") if callable(fmerge) and not isinstance(fmerge, _Fusion) else fmerge
    mode = getattr(obj, "mode", None)
    if mode not in ("sign_max", "max", "avg"):
        raise ValueError("unsupported fMerge %r: expected Maximum, Average or sign_max" % (fmerge,))
    return mode


# ---------------------------------------------------------------------------------------------------------
# model object
# ---------------------------------------------------------------------------------------------------------
class _BatchPipeline:
    """What tf.keras's `fit` does with a `keras.utils.Sequence` (OrderedEnqueuer: `workers=1`, `max_queue_size=10` are its defaults
    and the reference's call, nets/mj_uwyhNets_ba.py:963, leaves them alone): a background thread pulls the epoch's batches IN ORDER
    while the device trains, here additionally staging them in HBM -- pinned host copy, asynchronous transfer on a copy stream of its
    own, an event the training stream waits on -- so that neither the generator's host work nor the 35 MB of a 24-clip batch crossing
    PCIe sits between two steps.  One pipeline per epoch: the generator's `on_epoch_end` (the reference reshuffles there,
    data/mj_dataGeneratorMMUWYHsingle_repetitions.py) runs between two of them, never beside a fetch."""

    _END = object()

    STAGED_DEPTH = 4       # batches staged ahead in HBM at most (each holds pinned host memory too); max_queue_size beyond it buys nothing

    def __init__(self, gen, steps, device, depth, stage=True, ring=None, workers=1, flag_inputs=()):
        import queue
        import threading
        self.gen, self.steps, self.device, self.stage = gen, int(steps), device, stage
        # positions of the modality-flag inputs in X (GaitSetModel: the odd ones of a multimodal model): they stay on the host, where
        # GaitCore decides which masked (clip, modality) pairs to skip -- the model's own input split, not a guess from the shape
        self.flag_inputs = frozenset(int(i) for i in flag_inputs)
        self.workers = max(1, int(workers)) if hasattr(gen, "__len__") else 1     # (an iterator has one consumer)
        self.q = queue.Queue(maxsize=max(1, min(int(depth), self.STAGED_DEPTH) if stage else int(depth)))
        # pinned staging buffers: queue + the batch in the step + the one being filled; the caller keeps the list across epochs
        self.ring = ring if ring is not None else []
        while len(self.ring) < self.q.maxsize + 2:
            self.ring.append(dict())
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._run, name="ugaitnet-batches", daemon=True)
        self.thread.start()

    def _stage(self, arrays, stream, slot):
        """host arrays -> device tensors through the pinned buffers of ring slot `slot` (allocated once per shape: a fresh pinned
        allocation per batch costs more than the step itself), asynchronously on the copy stream"""
        import torch
        ring = self.ring[slot]
        if ring.get("event") is not None:
            ring["event"].synchronize()         # the transfer that last used these pinned buffers has finished
        outs = []
        with torch.cuda.stream(stream):
            for k, a in enumerate(arrays):
                if (isinstance(a, torch.Tensor) and a.is_cuda) or k in self.flag_inputs:
                    # (already in HBM; or a modality-flag column: it stays on the host -- from a device tensor the decision which
                    #  masked pairs to skip would synchronise with the previous step)
                    outs.append(a)
                    continue
                src = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                # pinned buffers are keyed by (input, shape): pipelines that alternate batch shapes on one ring (a last, smaller batch
                # of an epoch) never re-pin -- a fresh pinned allocation costs more than a step
                key = (k, tuple(src.shape))
                pin = ring.get(key)
                if pin is None:
                    pin = ring[key] = torch.empty(tuple(src.shape), dtype=torch.float32, pin_memory=True)
                # one plain memcpy on this thread (numpy releases the GIL for it; converts to fp32 on the way, as GaitCore._dev would).
                # NOT torch's copy_: its OpenMP team beside the training thread's launches cost 20 ms per batch on a 16-core share
                np.copyto(pin.numpy(), src, casting="same_kind")
                outs.append(pin.to(self.device, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record(stream)
        ring["event"] = ev
        return outs, ev

    def _run(self):
        import torch
        try:
            stream = None
            if self.stage:
                torch.cuda.set_device(self.device)
                stream = torch.cuda.Stream(self.device)
            sized = hasattr(self.gen, "__len__")
            pool, ahead = None, []
            if self.workers > 1:       # tf.keras's OrderedEnqueuer with workers > 1: `__getitem__` calls in parallel, handed on IN ORDER
                import concurrent.futures
                pool = concurrent.futures.ThreadPoolExecutor(self.workers, thread_name_prefix="ugaitnet-fetch")
                self.pool = pool
            nxt = 0
            for step in range(self.steps):
                if self.stop.is_set():
                    return
                if pool is not None:
                    while nxt < self.steps and len(ahead) < self.workers + self.q.maxsize:
                        ahead.append(pool.submit(self.gen.__getitem__, nxt % len(self.gen)))
                        nxt += 1
                    X, Y = ahead.pop(0).result()
                else:
                    X, Y = self.gen[step % len(self.gen)] if sized else next(self.gen)
                item = (X, Y, None)
                if self.stage:
                    staged, ev = self._stage(list(X) if isinstance(X, (list, tuple)) else [X], stream, step % len(self.ring))
                    item = (staged if isinstance(X, (list, tuple)) else staged[0], Y, ev)
                self._put(item)
            self._put(self._END)
        except BaseException as e:      # (surfaces in the training thread, at the step that would have used the batch)
            self._put(e)

    def _put(self, item):
        import queue
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def get(self):
        import torch
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        if item is self._END:
            raise StopIteration
        X, Y, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (X if isinstance(X, (list, tuple)) else [X]):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)       # (allocated under the copy stream: the allocator must not hand it out again before the step has read it)
        return X, Y

    def close(self):
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except Exception:
            pass
        # No timeout: the thread looks at `stop` every 0.1 s in _put and between two steps, so it ends after at most ONE fetch -- and
        # the caller's next moves (the generator's on_epoch_end reshuffle, the next epoch's pipeline on the same pinned ring) must never
        # run beside a fetch that is still in flight.  Fetches of the worker pool that already run are waited for as well.
        self.thread.join()
        if getattr(self, "pool", None) is not None:
            self.pool.shutdown(wait=True, cancel_futures=True)


class History:
    def __init__(self):
        self.epoch = []
        self.history = {}

    def _append(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class _Tensor:
    """Symbolic handle of a layer output (only what `Model(model.input, model.get_layer(n).output)` needs)."""

    def __init__(self, model, name):
        self.model, self.name = model, name


class LayerHandle:
    def __init__(self, model, name, units=None, weight_names=()):
        self._model, self.name, self.units, self._weight_names = model, name, units, tuple(weight_names)
        self.trainable = True
        self.output = _Tensor(model, name)

    def get_weights(self):
        return [self._model.core.store.get(n) for n in self._weight_names]

    def set_weights(self, arrays):
        for n, a in zip(self._weight_names, arrays):
            self._model.core.store.set(n, a)
        self._model.core.weights_changed()


_MOD_NAMES = ("of", "gray", "depth")


class TripletHardLoss:
    """Stand-in for tfa.losses.TripletHardLoss(margin) in `model.compile(loss=[...])`: selects the batch-hard kernel
    (ugn_triplet_hard_fwd_bwd; soft=False, L2 distances, per bin of the [62,B,256] signature)."""

    def __init__(self, margin=1.0, soft=False, distance_metric="L2", name=None):
        if soft or distance_metric != "L2":
            raise NotImplementedError("TripletHardLoss on the MI355X path: soft=False and distance_metric='L2' only")
        self.margin = float(margin)


class GaitSetModel:
    """What `UWYHSemiNet.build(..., gaitset=True)` returns: a compiled model driving the HIP engine."""

    dtype = "float32"

    def __init__(self, input_shapes, nclasses, loss_weights, margin, optimizer, fmerge, multimodal, seed=None,
                 name="ugaitnet"):
        from .engine import GaitCore
        self.name = name
        self.input_shapes = [tuple(s) for s in input_shapes]
        for s in self.input_shapes:
            if len(s) != 4 or tuple(s[1:3]) != (60, 60) or s[3] not in (1, 2):
                raise ValueError("gaitset branch expects inputs of shape (L, 60, 60, 1|2), got %r" % (s,))
        self.nclasses = int(nclasses)
        self.multimodal = bool(multimodal)
        self.margin = float(margin)
        self.fmerge_mode = fusion_mode(fmerge) if multimodal else "max"
        if not isinstance(optimizer, Adam):
            raise NotImplementedError("only optimizers.Adam is implemented on the MI355X path (the reference's "
                                      "published configurations all use --optimizer=Adam); got %r" % (optimizer,))
        self.optimizer = optimizer
        lw = list(loss_weights) if isinstance(loss_weights, (list, tuple, np.ndarray)) else [1.0]
        if self.nclasses == 0:
            lw = [1.0]
        self.loss_weights = [float(v) for v in lw]
        self.loss = ["triplet_loss(margin=%g)" % self.margin] + (["categorical_crossentropy"] if self.nclasses else [])
        self.sig_name = "signature" if self.multimodal else "mat_mul"
        import torch
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        self.core = GaitCore([s[3] for s in self.input_shapes], nclasses=self.nclasses, multimodal=self.multimodal,
                             fuse_mode=self.fmerge_mode, margin=self.margin,
                             loss_weights=(self.loss_weights[0], self.loss_weights[1] if len(self.loss_weights) > 1 else 0.0),
                             seed=seed, lr=optimizer.lr, beta_1=optimizer.beta_1, beta_2=optimizer.beta_2,
                             epsilon=optimizer.epsilon, world_size=world,
                             dp_mode=os.environ.get("UGN_DP_MODE", "replica"),
                             # a masked (clip, modality) pair is multiplied by 0 in the gate (nets/mj_uwyhNets_ba.py:51-54): its
                             # encoder work is skipped -- same outputs and gradients, 1.5x the clips/s on the 7-pattern masks
                             # (bench.py's headline computes those pairs; `value_skip_masked` is this mode).  UGN_SKIP_MASKED=0: dense
                             skip_masked=os.environ.get("UGN_SKIP_MASKED", "1") != "0",
                             conv_precision=None)      # engine.DEFAULT_PRECISION (UGN_CONV_PRECISION; "f32x3")
        if world > 1:  # replicas start from identical weights (MirroredStrategy semantics)
            self.core.join_pack()
            torch.distributed.broadcast(self.core.store.flat, src=0)
            self.core.weights_changed()
        self.input_names = []
        mods = _MOD_NAMES[:len(self.input_shapes)]
        for m in mods:
            self.input_names.append(m + "input1")
            if self.multimodal:
                self.input_names.append(m + "use1")
        self.input = [_Tensor(self, n) for n in self.input_names]
        self.inputs = self.input
        self.layers = []
        for mi, m in enumerate(mods):
            for n in ("a1", "a2", "b1", "b2", "a3", "a4", "b3", "b4", "a5", "a6"):
                self.layers.append(LayerHandle(self, "%sBranch_%s" % (m, n), weight_names=["m%d.%s" % (mi, n)]))
            self.layers.append(LayerHandle(self, "%sBranch_mat_mul" % m, weight_names=["m%d.fc" % mi]))
            if self.multimodal:
                self.layers.append(LayerHandle(self, "gate_%s1" % m))
        if self.multimodal:
            self.layers += [LayerHandle(self, "fusion"), LayerHandle(self, "signature")]
        if self.nclasses:
            self.layers += [LayerHandle(self, "flatten"),
                            LayerHandle(self, "classprob", units=self.nclasses, weight_names=["head.wc", "head.bc"])]
        self.stop_training = False

    # ---- Keras-like surface ------------------------------------------------------------------------------
    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: %s" % name)

    def compile(self, optimizer=None, loss=None, loss_weights=None, metrics=None):
        # `loss`: the first entry may be a TripletHardLoss marker (compile_hard, nets/mj_uwyhNets_ba.py:1301-1306) or the
        # reference's triplet_loss(margin) closure; anything carrying `.margin` updates the margin
        first = loss[0] if isinstance(loss, (list, tuple)) and loss else loss
        if first is not None:
            if getattr(first, "margin", None) is not None:
                self.margin = self.core.margin = float(first.margin)
            hard = isinstance(first, TripletHardLoss)
            self.core.triplet_mode = "hard" if hard else "all"
            self.loss[0] = ("TripletHardLoss(margin=%g)" if hard else "triplet_loss(margin=%g)") % self.margin
        if optimizer is not None:
            if not isinstance(optimizer, Adam):
                raise NotImplementedError("only optimizers.Adam is implemented")
            self.optimizer = optimizer
        if loss_weights is not None:
            lw = [float(v) for v in (loss_weights if isinstance(loss_weights, (list, tuple)) else [loss_weights])]
            self.loss_weights = lw
            self.core.loss_weights = (lw[0], lw[1] if len(lw) > 1 else 0.0)

    def count_params(self):
        return int(sum(int(np.prod(s)) for s in self.core.store.shapes.values()))

    def summary(self, print_fn=print):
        print_fn('Model: "%s" (MI355X HIP engine; %d modality branch(es), fusion=%s)' %
                 (self.name, len(self.input_shapes), self.fmerge_mode if self.multimodal else "none"))
        for n in self.core.store.names:
            print_fn("  %-12s %s" % (n, self.core.store.shapes[n]))
        print_fn("Total params: %d" % self.count_params())

    def _split_x(self, X):
        nm = len(self.input_shapes)
        if not self.multimodal:
            x = X[0] if isinstance(X, (list, tuple)) else X
            return [x], None
        if len(X) != 2 * nm:
            raise ValueError("expected %d inputs %s, got %d" % (2 * nm, self.input_names, len(X)))
        return [X[2 * i] for i in range(nm)], [X[2 * i + 1] for i in range(nm)]

    def _flag_inputs(self):
        """positions of the modality-flag inputs in X (the `use` columns: odd positions of a multimodal model, none otherwise)"""
        return tuple(range(1, 2 * len(self.input_shapes), 2)) if self.multimodal else ()

    def _split_y(self, y):
        if self.nclasses:
            labels, onehot = y
            return np.asarray(labels).reshape(-1), onehot
        labels = y[0] if isinstance(y, (list, tuple)) else y
        return np.asarray(labels).reshape(-1), None

    def _sync_lr(self):
        c, o = self.core, self.optimizer
        c.lr, c.beta_1, c.beta_2, c.epsilon = float(o.lr), o.beta_1, o.beta_2, o.epsilon

    def train_on_batch(self, X, y):
        xs, uses = self._split_x(X)
        labels, onehot = self._split_y(y)
        self._sync_lr()
        self.core.train_step(xs, uses, labels, onehot)
        return self._logs(self.core.losses())

    def test_on_batch(self, X, y):
        xs, uses = self._split_x(X)
        labels, onehot = self._split_y(y)
        self.core.forward_loss_only(xs, uses, labels, onehot)
        return self._logs(self.core.losses())

    def _logs(self, ls):
        out = {"loss": ls["loss"]}
        if self.nclasses:
            out[self.sig_name + "_loss"] = ls["triplet"]
            out["classprob_loss"] = ls["xent"]
            out["classprob_acc"] = ls["acc"]
        return out

    def predict(self, X, batch_size=None, verbose=0):
        """Outputs of the compiled graph: [signature [62,B,256], classprob [B,ncls]] (or the signature alone)."""
        xs, uses = self._split_x(X)
        with self.core.arithmetic(_infer_precision()):
            sig, _, probs = self.core.predict(xs, uses)
        sig = sig.cpu().numpy()
        return [sig, probs.cpu().numpy()] if self.nclasses else sig

    def predict_layer(self, name, X):
        xs, uses = self._split_x(X)
        with self.core.arithmetic(_infer_precision()):
            sig, flat, probs = self.core.predict(xs, uses)
        if name == "flatten":
            return flat.cpu().numpy()
        if name in ("signature", "mat_mul", self.sig_name):
            return sig.cpu().numpy()
        if name == "classprob":
            return probs.cpu().numpy()
        if name == "fusion" and self.multimodal:
            return self.core.fused.cpu().numpy()
        raise ValueError("no output tap for layer %r" % name)

    def evaluate(self, generator, steps=None, verbose=0, workers=1, max_queue_size=10):
        """Mean of the per-batch losses / metrics over `steps` batches; the batches are fetched and staged by the same background
        pipeline as in `fit` (workers=0: on this thread)."""
        n = len(generator) if steps is None else steps
        acc = {}
        if not hasattr(self, "_pinned_ring_eval"):       # (a ring of its own: validation inside `fit` alternates with the training pipeline)
            self._pinned_ring_eval = []
        feed = _BatchPipeline(generator, n, self.core.device, max_queue_size, ring=self._pinned_ring_eval, workers=workers,
                              flag_inputs=self._flag_inputs()) if workers and workers > 0 else None
        try:
            for i in range(n):
                X, y = feed.get() if feed is not None else generator[i % len(generator)]
                for k, v in self.test_on_batch(X, y).items():
                    acc[k] = acc.get(k, 0.0) + v / n
        finally:
            if feed is not None:
                feed.close()
        return acc

    def fit(self, x=None, y=None, validation_data=None, epochs=1, steps_per_epoch=None, callbacks=None,
            validation_steps=None, initial_epoch=0, verbose=2, workers=1, max_queue_size=10, use_multiprocessing=False,
            pipeline=None, **kwargs):
        """The loop `model.fit(training_generator, ...)` runs in nets/mj_uwyhNets_ba.py:963: per epoch
        steps_per_epoch batches from the keras.utils.Sequence-like generator, Keras callbacks, History.

        workers / max_queue_size: as in tf.keras (defaults 1 / 10): a background thread fetches the epoch's batches in order and
        stages them in HBM while the device trains (`_BatchPipeline`); workers > 1 call the Sequence's `__getitem__` in parallel and
        hand the batches on in order; workers=0 fetches on the training thread, as Keras does.
        pipeline (default: on, unless a callback listens to batch events): step k's losses are read after step k + 1 has been
        queued, so the device never waits for the host between two steps; the History, the epoch logs and every parameter are
        bit-identical to the synchronous loop (tests/test_api_gpu.py::test_fit_pipeline_matches_the_synchronous_loop) -- only
        `on_batch_end` would see the model one step ahead, hence the default.  use_multiprocessing is accepted and ignored (the
        reference passes its default)."""
        gen = x
        hist = History()
        callbacks = list(callbacks or [])
        batch_hooks = ("on_batch_begin", "on_batch_end", "on_train_batch_begin", "on_train_batch_end")
        if pipeline is None:
            pipeline = not any(_listens(cb, h) for cb in callbacks for h in batch_hooks)
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        _call(callbacks, "on_train_begin", {})
        self.stop_training = False
        for epoch in range(initial_epoch, epochs):
            _call(callbacks, "on_epoch_begin", epoch, {})
            n = steps_per_epoch if steps_per_epoch else len(gen)
            sums = {}

            def account(step, logs):
                for k, v in logs.items():
                    sums[k] = sums.get(k, 0.0) + v
                _call(callbacks, "on_batch_end", step, logs)
            if not hasattr(self, "_pinned_ring"):
                self._pinned_ring = []
            feed = _BatchPipeline(gen, n, self.core.device, max_queue_size, ring=self._pinned_ring, workers=workers,
                                  flag_inputs=self._flag_inputs()) if workers and workers > 0 else None
            try:
                pending = None
                for step in range(n):
                    if feed is not None:
                        X, Y = feed.get()
                    else:
                        X, Y = gen[step % len(gen)] if hasattr(gen, "__len__") else next(gen)
                    if not pipeline:
                        account(step, self.train_on_batch(X, Y))
                        continue
                    xs, uses = self._split_x(X)
                    labels, onehot = self._split_y(Y)
                    self._sync_lr()
                    self.core.train_step(xs, uses, labels, onehot)
                    queued = self.core.losses_async()
                    if pending is not None:
                        account(step - 1, self._logs(pending.result()))
                    pending = queued
                if pending is not None:
                    account(n - 1, self._logs(pending.result()))
            finally:
                if feed is not None:
                    feed.close()
            logs = {k: v / n for k, v in sums.items()}
            if validation_data is not None:
                vn = validation_steps if validation_steps else len(validation_data)
                for k, v in self.evaluate(validation_data, vn, workers=workers, max_queue_size=max_queue_size).items():
                    logs["val_" + k] = v
            logs["lr"] = float(self.optimizer.lr)
            hist._append(epoch, logs)
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in logs.items()), flush=True)
            if hasattr(gen, "on_epoch_end"):
                gen.on_epoch_end()
            _call(callbacks, "on_epoch_end", epoch, logs)
            if self.stop_training:
                break
        _call(callbacks, "on_train_end", {})
        return hist

    # ---- persistence (npz container; h5py/deepdish are not available on the target image) --------------------
    def get_config(self):
        return dict(input_shapes=[list(s) for s in self.input_shapes], nclasses=self.nclasses,
                    loss_weights=self.loss_weights, margin=self.margin, fmerge=self.fmerge_mode,
                    multimodal=self.multimodal, optimizer=self.optimizer.get_config())

    def _weights_dict(self):
        return {"w/" + n: self.core.store.get(n) for n in self.core.store.names}

    def _params(self):
        return {n: self.core.store.get(n) for n in self.core.store.names}

    def save_weights(self, path):
        """`*.h5` / `*.hdf5`: the Keras HDF5 weights layout under the layer names the reference's graph carries
        (ugaitnet_amd/keras_h5.py), readable by its `load_weights(by_name=True)`; any other name: an npz container."""
        if _is_h5_name(path):
            from . import keras_h5
            keras_h5.write_weights(os.fspath(path), self._params(), self.core.in_channels, self.nclasses)
        else:
            _savez(path, self._weights_dict())

    def save(self, path):
        """Weights + configuration + optimizer state (`model.save`).  HDF5 names: the weights below /model_weights as Keras
        lays them out, this build's configuration in the root attribute `ugaitnet_config`, Adam's state below
        /optimizer_weights."""
        opt = {"m": self.core.store.m.cpu().numpy(), "v": self.core.store.v.cpu().numpy(),
               "iterations": np.array([self.core.iterations], dtype=np.int64)}
        if _is_h5_name(path):
            from . import keras_h5
            extra = {"optimizer_weights/" + k: v for k, v in opt.items()}
            extra["ugaitnet_config"] = json.dumps(self.get_config())
            keras_h5.write_weights(os.fspath(path), self._params(), self.core.in_channels, self.nclasses, extra=extra,
                                   below="model_weights")
            return
        d = self._weights_dict()
        d["config_json"] = np.frombuffer(json.dumps(self.get_config()).encode(), dtype=np.uint8)
        d.update({"opt/" + k: v for k, v in opt.items()})
        _savez(path, d)

    def _set_opt_state(self, m, v, iterations):
        import torch
        if m.shape[0] == self.core.store.numel:
            self.core.store.m.copy_(torch.from_numpy(np.ascontiguousarray(m)))
            self.core.store.v.copy_(torch.from_numpy(np.ascontiguousarray(v)))
            self.core.iterations = int(np.asarray(iterations).reshape(-1)[0])

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        """Keras semantics: by_name=False wants every weight of the model in the file; by_name=True takes what is there;
        skip_mismatch=True (by_name only in Keras) leaves a weight of a different shape untouched instead of failing.
        Reads Keras HDF5 weight / model files (the reference's checkpoints) and this build's npz containers."""
        shapes = self.core.store.shapes
        if _file_is_h5(path):
            from . import h5lite, keras_h5
            found = keras_h5.assign(keras_h5.read_layers(os.fspath(path)), len(self.input_shapes), self.nclasses)
            f = h5lite.File(os.fspath(path))
            opt = None
            if "optimizer_weights" in f and "m" in f["optimizer_weights"]:
                g = f["optimizer_weights"]
                opt = (g["m"].read(), g["v"].read(), g["iterations"].read())
        else:
            with np.load(path, allow_pickle=False) as z:
                found = {k[2:]: z[k] for k in z.files if k.startswith("w/")}
                opt = (z["opt/m"], z["opt/v"], z["opt/iterations"]) if "opt/m" in z.files else None
        for n in self.core.store.names:
            a = found.get(n)
            if a is None:
                if by_name:
                    continue
                raise ValueError("weight %s missing in %s" % (n, path))
            if tuple(a.shape) != tuple(shapes[n]):
                if skip_mismatch:
                    continue
                raise ValueError("shape mismatch for %s: %r vs %r" % (n, a.shape, shapes[n]))
            self.core.store.set(n, a)
        if opt is not None:
            self._set_opt_state(*opt)
        self.core.weights_changed()


def _is_h5_name(path):
    return os.path.splitext(os.fspath(path))[1].lower() in (".h5", ".hdf5", ".hdf")


def _file_is_h5(path):
    with open(os.fspath(path), "rb") as fh:
        return fh.read(8) == b"\x89HDF\r\n\x1a\n"


def _savez(path, arrays):
    """npz container written to EXACTLY `path` (the mains name their files *.hdf5; np.load does not mind)."""
    with open(os.fspath(path), "wb") as f:
        np.savez(f, **arrays)


def _listens(cb, hook):
    """does this callback define `hook` itself (a no-op inherited from a base class called Callback does not count)"""
    if not callable(getattr(cb, hook, None)):
        return False
    for klass in type(cb).__mro__:
        if hook in vars(klass):
            return klass.__name__ != "Callback"
    return True


def _call(callbacks, method, *args):
    for cb in callbacks:
        fn = getattr(cb, method, None)
        if fn is not None:
            fn(*args)


def _infer_precision():
    """engine.DEFAULTS.infer_precision ('f32': inference in IEEE fp32 whatever the training arithmetic; 'same': the model's own)"""
    from . import engine
    return None if engine.DEFAULTS.infer_precision == "same" else engine.DEFAULTS.infer_precision


class SubModel:
    """`Model(model.input, model.get_layer('flatten').output)`: a forward-only tap (test mains, :139-148)."""

    def __init__(self, model, layer_name):
        self.model, self.layer_name = model, layer_name

    def predict(self, X, batch_size=None, verbose=0):
        return self.model.predict_layer(self.layer_name, X)

    __call__ = predict


def Model(inputs=None, outputs=None):
    if isinstance(outputs, _Tensor):
        return SubModel(outputs.model, outputs.name)
    raise TypeError("Model(inputs, outputs): outputs must be `model.get_layer(name).output` of a GaitSetModel")


def load_model(path, custom_objects=None, compile=False):
    """Counterpart of keras load_model for files written by GaitSetModel.save() (HDF5 or npz).  A model file written by
    Keras itself carries a Keras graph description this build cannot execute: rebuild the model with `build_or_load` and
    restore its arrays with `load_weights(path, by_name=True)`, as the reference's own restore path does (:610-630)."""
    if _file_is_h5(path):
        from . import h5lite
        attrs = h5lite.File(os.fspath(path)).attrs
        if "ugaitnet_config" not in attrs:
            raise ValueError("%s holds no ugaitnet_config attribute (a Keras-written model file?): build the model with "
                             "build_or_load(...) and call load_weights(path, by_name=True)" % path)
        c = attrs["ugaitnet_config"]
        cfg = json.loads(c.decode() if isinstance(c, bytes) else str(c))
    else:
        with np.load(path, allow_pickle=False) as z:
            cfg = json.loads(bytes(z["config_json"]).decode())
    fm = {"sign_max": sign_max, "max": Maximum, "avg": Average}[cfg["fmerge"]]
    oc = cfg["optimizer"]
    opt = Adam(lr=oc["lr"], beta_1=oc["beta_1"], beta_2=oc["beta_2"], epsilon=oc["epsilon"])
    m = GaitSetModel(cfg["input_shapes"], cfg["nclasses"], cfg["loss_weights"], cfg["margin"], opt, fm, cfg["multimodal"])
    m.load_weights(path)
    return m
