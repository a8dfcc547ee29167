"""Drop-in counterpart of the reference's nets/mj_uwyhNets_ba.py for the gaitset path (gaitset=True).

Same class / static-method names, argument order, defaults and error behaviour as the reference
(nets/mj_uwyhNets_ba.py:581-999 `UWYHSemiNet`, :1003-1397 `UWYHSemiNet3Mods`); the returned model object exposes the
Keras subset the mains use (ugaitnet_amd.keras_compat.GaitSetModel) and runs every arithmetic op as a HIP kernel on
MI355X.  Only the gaitset branch is implemented: the legacy 2-D / 3-D branches are out of scope (SURVEY.md section 2,
rows 9-13) and raise NotImplementedError instead of silently building something else.

On the gaitset path the reference ignores number_convolutional_layers / filters_size / filters_numbers / ndense_units /
dropout / weight_decay (the branch is hard-coded, :419-484); they are accepted and ignored here too.
"""
from __future__ import annotations

import os.path as osp

import numpy as np

from ..keras_compat import (Adam, Average, GaitSetModel, Maximum, Model, SGD, load_model, optimizers,  # noqa: F401
                            sign_max)
from .triplet_loss_all import triplet_loss  # noqa: F401


class MatMul:
    """Descriptor of the reference's MatMul layer (:23-48): kernel [bin_num*2, 128, hidden_dim], GlorotUniform,
    applied as tf.matmul(x, kernel) per bin.  The computation is ugn_binfc_fwd / ugn_binfc_bwd."""

    def __init__(self, bin_num=31, hidden_dim=256, **kwargs):
        self.bin_num, self.hidden_dim = bin_num, hidden_dim

    def get_config(self):
        return {"bin_num": self.bin_num, "hidden_dim": self.hidden_dim}


def mj_tensor_times_scalar(d):
    """Reference :51-54 (gate).  Host helper for numpy arrays; inside the model the gate is fused into
    ugn_gate_fuse_fwd."""
    tensor, scalar = d[0], d[1]
    return tensor * scalar


def _require_gaitset(gaitset, use3D, aux_losses, smoothlabels, init_branches, ndense_units):
    if not gaitset:
        raise NotImplementedError("only the gaitset=True branch is implemented on MI355X; the legacy 2-D/3-D branches "
                                  "of nets/mj_uwyhNets_ba.py:66-417 are out of scope")
    if use3D:
        raise NotImplementedError("use3D is not part of the gaitset hot path")
    if aux_losses:
        raise NotImplementedError("aux_losses (per-modality classifiers) are not implemented")
    if smoothlabels:
        raise NotImplementedError("label smoothing is not implemented")
    if init_branches is not None and any(v for v in init_branches.values()):
        raise NotImplementedError("init_branches (pre-trained Keras branches) cannot be loaded: no HDF5 reader here")
    if isinstance(ndense_units, (list, tuple)) and len(ndense_units) > 1:
        raise NotImplementedError("the extra dense 'code' layer is not on the gaitset path the mains run (ndense=0)")


def _freeze_like_reference(model, build, initnet, freeze_convs, freeze_all, weights_filename):
    """The gaitset arm of reference nets/mj_uwyhNets_ba.py:635-649 (and :1375-1389 for three modalities): with freeze_convs
    or freeze_all a FRESH model is built and the `<initnet>_weights.hdf5` file loaded into it by name; freeze_all then sets
    trainable = False on every layer but the last one (`classprob`) -- freeze_convs alone freezes nothing on this path."""
    if not (freeze_convs or freeze_all):
        return model
    import os
    model = build()
    filewes = weights_filename(initnet)
    model.load_weights(filewes if os.path.exists(filewes) else initnet, by_name=True, skip_mismatch=True)
    if freeze_all:
        model.core.frozen_branches = True
    return model


def _resolve_config(netconfig):
    """A configuration as stored in `model-config.hdf5` (ugaitnet_amd.ddconfig) names its optimizer and merge function
    (`'optimizer': 'Adam'`, `'fMerge': 'Maximum'`: strings, mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:474-489); the
    surgery route then overwrites both with live objects.  Either spelling is accepted."""
    cfg = dict(netconfig)
    opt = cfg.get("optimizer")
    if isinstance(opt, str):
        table = {"adam": Adam, "sgd": SGD}
        if opt.lower() not in table:
            raise ValueError("model configuration: unknown optimizer %r (Adam, SGD)" % opt)
        cfg["optimizer"] = table[opt.lower()]()        # Keras' defaults, as compile(optimizer='Adam') takes them
    fm = cfg.get("fMerge", Maximum)
    if isinstance(fm, str):
        table = {"Maximum": Maximum, "Average": Average, "sign_max": sign_max}
        if fm not in table:
            raise ValueError("model configuration: unknown fMerge %r (Maximum, Average, sign_max)" % fm)
        fm = table[fm]
    cfg["fMerge"] = fm
    shp = cfg["input_shape"]
    cfg["input_shape"] = [tuple(int(v) for v in t) for t in shp] if isinstance(shp[0], (list, tuple)) else tuple(int(v) for v in shp)
    return cfg


def _load_netconfig(modelpath):
    """The dictionary of `<directory of modelpath>/model-config.hdf5`, or None when there is no such file."""
    import os
    from .. import ddconfig
    fconfig = UWYHSemiNet.get_netconfig_filename(modelpath)
    return ddconfig.load(fconfig) if os.path.exists(fconfig) else None


def _surgery(cls, initnet, build, overrides):
    """Reference :610-630 / :1331-1352: a checkpoint whose classification layer has another width is rebuilt from the stored
    configuration (with the caller's loss / optimizer settings written over it) and takes the compatible weights of
    `<initnet>_weights.hdf5` by name.  Without a configuration file the model is built from the caller's arguments."""
    import os
    netconfig = _load_netconfig(initnet)
    if netconfig is None:
        model = build()
        model.load_weights(initnet, by_name=True, skip_mismatch=True)
        return model
    netconfig.update(overrides)
    model = cls.build_by_config(netconfig)
    filewes = UWYHSemiNet.get_weights_filename(initnet)
    model.load_weights(filewes if os.path.exists(filewes) else initnet, by_name=True, skip_mismatch=True)
    return model


class UWYHSemiNet:
    """1- or 2-modality model (reference :581-999).  `input_shapes`: a tuple (L,60,60,C) for one modality, a list of two
    such tuples for two (first = optical flow, second = gray)."""

    @staticmethod
    def build(input_shapes, number_convolutional_layers, filters_size, filters_numbers, ndense_units=512,
              weight_decay=1e-4, dropout=0.4, optimizer=None, margin=0.2, nclasses=0, loss_weights=[1.0, 1.0],
              use3D=False, smoothlabels=0, postriplet=1, init_branches=None, freeze_branches=False, aux_losses=False,
              fMerge=Maximum, fActivation='relu', alpha=0.3, gaitset=False, seed=None):
        _require_gaitset(gaitset, use3D, aux_losses, smoothlabels, init_branches, ndense_units)
        if number_convolutional_layers < 1:
            print("ERROR: Number of convolutional layers must be greater than 0")
        multimodal = type(input_shapes) is list          # reference :693-698 (`type(input_shapes) is list`)
        shapes = list(input_shapes) if multimodal else [tuple(input_shapes)]
        if multimodal and len(shapes) != 2:
            raise ValueError("UWYHSemiNet takes one input shape (tuple) or a list of two; use UWYHSemiNet3Mods for three")
        optimizer = optimizers.SGD(0.001, 0.9) if optimizer is None else optimizer
        return GaitSetModel(shapes, nclasses, loss_weights, margin, optimizer, fMerge, multimodal, seed=seed)

    @staticmethod
    def build_by_config(netconfig):
        """Reference :299-340 (`mj_buildnet_by_config`): the model a stored configuration describes.  A configuration of the
        gaitset mains carries no `gaitset` key (it is a command-line switch there); this build has no other path."""
        c = _resolve_config(netconfig)
        return UWYHSemiNet.build(c["input_shape"], len(c["filters_numbers"]), c["filters_size"], c["filters_numbers"],
                                 c["ndense_units"], c["weight_decay"], c["dropout"], nclasses=c.get("nclasses", 150),
                                 loss_weights=c.get("loss_weights", [1.0, 0.1]), optimizer=c["optimizer"],
                                 margin=float(c["margin"]), use3D=bool(c.get("use3D", False)),
                                 postriplet=c.get("postriplet", 1), fMerge=c["fMerge"],
                                 fActivation=c.get("fActivation", "relu"), gaitset=c.get("gaitset", True))

    @staticmethod
    def get_weights_filename(modelpath):
        bdir, bname = osp.dirname(modelpath), osp.basename(modelpath)
        return osp.join(bdir, osp.splitext(bname)[0] + "_weights.hdf5")

    @staticmethod
    def get_netconfig_filename(modelpath):
        return osp.join(osp.dirname(modelpath), "model-config.hdf5")

    @staticmethod
    def loadnet(netpath: str):
        """Reference :553-577 (and the live fallback of :1008-1030): a model file this build wrote is loaded as such; one it
        cannot interpret (a Keras-written model) is rebuilt from `model-config.hdf5` beside it, when that exists, and takes
        the arrays of `<netpath>_weights.hdf5` by name."""
        print(netpath)
        try:
            return load_model(netpath, compile=False)
        except ValueError:
            netconfig = _load_netconfig(netpath)
            if netconfig is None:
                raise
            import os
            shapes = netconfig["input_shape"]
            cls = UWYHSemiNet3Mods if isinstance(shapes[0], (list, tuple)) and len(shapes) == 3 else UWYHSemiNet
            model = cls.build_by_config(netconfig)
            filewes = UWYHSemiNet.get_weights_filename(netpath)
            model.load_weights(filewes if os.path.exists(filewes) else netpath, by_name=True)
            return model

    @staticmethod
    def build_or_load(input_shapes, number_convolutional_layers, filters_size, filters_numbers, ndense_units=512,
                      weight_decay=1e-4, dropout=0.4, optimizer=None, margin=0.2, nclasses=0, loss_weights=[1.0, 1.0],
                      initnet="", freeze_convs=False, use3D=False, smoothlabels=0, freeze_all=False, postriplet=1,
                      init_branches=None, freeze_branches=False, aux_losses=False, fMerge=Maximum, fActivation='relu',
                      gaitset=False, seed=None):
        if gaitset:
            fActivation = 'leaky'
        if freeze_branches:
            raise NotImplementedError("freeze_branches (with init_branches) is not implemented on the MI355X path")
        build = lambda: UWYHSemiNet.build(input_shapes, number_convolutional_layers, filters_size, filters_numbers,
                                          ndense_units, weight_decay, dropout, optimizer, margin, nclasses, loss_weights,
                                          use3D=use3D, smoothlabels=smoothlabels, postriplet=postriplet,
                                          init_branches=init_branches, freeze_branches=freeze_branches,
                                          aux_losses=aux_losses, fMerge=fMerge, fActivation=fActivation, gaitset=gaitset,
                                          seed=seed)
        if initnet == "":
            model = build()
        else:
            model_base = UWYHSemiNet.loadnet(initnet)
            try:
                units = model_base.get_layer("classprob").units
            except ValueError:
                print("This model doesn't contain classification layer")
                units = nclasses
            if units != nclasses:
                print("Surgery needed: {} vs {}".format(units, nclasses))
                model = _surgery(UWYHSemiNet, initnet, build,
                                 dict(nclasses=nclasses, loss_weights=loss_weights, dropout=dropout, margin=margin,
                                      optimizer=optimizer if optimizer is not None else optimizers.SGD(0.001, 0.9), use3D=use3D,
                                      postriplet=postriplet, fMerge=fMerge, fActivation=fActivation, gaitset=gaitset))
            else:
                model = model_base
            model = _freeze_like_reference(model, build, initnet, freeze_convs, freeze_all, UWYHSemiNet.get_weights_filename)
        print("Alright")
        return model

    @staticmethod
    def fit_generator(model, epochs, callbacks, training_generator, validation_generator, current_step, steps_per_epoch,
                      validation_steps, nworkers=0, new_lr=None):
        """Reference :937-968.  Returns (model, hist)."""
        if new_lr is not None:
            model.optimizer.lr = float(new_lr)
            print("INFO: learning rate has been changed to {}".format(new_lr))
        hist = model.fit(training_generator, validation_data=validation_generator, epochs=epochs,
                         steps_per_epoch=steps_per_epoch, callbacks=callbacks, validation_steps=validation_steps,
                         initial_epoch=current_step, verbose=2)
        return model, hist

    @staticmethod
    def encode(model, batch_data, use_data, gaitset=False):
        """Reference :970-999: per-branch codes, gate, Maximum, l2_normalize(axis=1), numpy.  For gaitset models the
        reference looks up a second Flatten layer that its own graph never creates; the
        behaviour it intends -- gated branch outputs, elementwise Maximum, batch-axis normalisation -- is what runs here,
        on the GPU, through the model's own kernels with fusion mode 'max'."""
        core = model.core
        xs = [np.asarray(b) for b in batch_data[:core.nmod]]
        uses = [np.asarray(u, dtype=np.float32).reshape(-1, 1) for u in use_data[:core.nmod]]
        saved = core.fuse_mode
        core.fuse_mode = "max"
        try:
            from ..keras_compat import _infer_precision
            with core.arithmetic(_infer_precision()):       # (IEEE fp32 unless UGN_INFER_PRECISION=same: engine.GaitCore.arithmetic)
                sig = core.forward(xs, uses)
            return sig.cpu().numpy()
        finally:
            core.fuse_mode = saved


class UWYHSemiNet3Mods(UWYHSemiNet):
    """3-modality model (reference :1003-1397); inputs ofinput1, ofuse1, grayinput1, grayuse1, depthinput1, depthuse1."""

    @staticmethod
    def build(input_shapes, number_convolutional_layers, filters_size, filters_numbers, ndense_units=512,
              weight_decay=1e-4, dropout=0.4, optimizer=None, margin=0.2, nclasses=0, loss_weights=[1.0, 1.0],
              use3D=False, smoothlabels=0, postriplet=1, init_branches=None, freeze_branches=False, aux_losses=False,
              fMerge=Maximum, normbfmerge=False, fActivation='relu', alpha=0.3, gaitset=False, seed=None):
        _require_gaitset(gaitset, use3D, aux_losses, smoothlabels, init_branches, ndense_units)
        if number_convolutional_layers < 1:
            print("ERROR: Number of convolutional layers must be greater than 0")
        shapes = list(input_shapes)
        if len(shapes) != 3:
            raise ValueError("UWYHSemiNet3Mods needs three input shapes (of, gray, depth)")
        optimizer = optimizers.SGD(0.001, 0.9) if optimizer is None else optimizer
        return GaitSetModel(shapes, nclasses, loss_weights, margin, optimizer, fMerge, True, seed=seed)

    @staticmethod
    def build_by_config(netconfig):
        """Reference :299-340 with `UWYHSemiNet3Mods.build` (:1019-1023)."""
        c = _resolve_config(netconfig)
        return UWYHSemiNet3Mods.build(c["input_shape"], len(c["filters_numbers"]), c["filters_size"], c["filters_numbers"],
                                      c["ndense_units"], c["weight_decay"], c["dropout"], nclasses=c.get("nclasses", 150),
                                      loss_weights=c.get("loss_weights", [1.0, 0.1]), optimizer=c["optimizer"],
                                      margin=float(c["margin"]), use3D=bool(c.get("use3D", False)),
                                      postriplet=c.get("postriplet", 1), fMerge=c["fMerge"],
                                      fActivation=c.get("fActivation", "relu"), gaitset=c.get("gaitset", True))

    @staticmethod
    def compile_hard(model, optimizer, loss_weights, margin):
        """Reference :1301-1306: recompile with tfa.losses.TripletHardLoss(margin) in place of the batch-all loss."""
        from ..keras_compat import TripletHardLoss
        model.compile(optimizer=optimizer, loss=[TripletHardLoss(margin=margin)] + list(model.loss[1:]), loss_weights=loss_weights,
                      metrics=[[], 'acc'])
        return model

    @staticmethod
    def build_or_load(input_shapes, number_convolutional_layers, filters_size, filters_numbers, ndense_units=512,
                      weight_decay=1e-4, dropout=0.4, optimizer=None, margin=0.2, nclasses=0, loss_weights=[1.0, 1.0],
                      initnet="", freeze_convs=False, use3D=False, smoothlabels=0, freeze_all=False, postriplet=1,
                      init_branches=None, freeze_branches=False, aux_losses=False, fMerge=Maximum, normbfmerge=False,
                      fActivation='relu', alpha=0.3, gaitset=False, seed=None):
        if gaitset:
            fActivation = 'leaky'
        if freeze_branches:
            raise NotImplementedError("freeze_branches (with init_branches) is not implemented on the MI355X path")
        build = lambda: UWYHSemiNet3Mods.build(input_shapes, number_convolutional_layers, filters_size, filters_numbers,
                                               ndense_units, weight_decay, dropout, optimizer, margin, nclasses,
                                               loss_weights, use3D=use3D, smoothlabels=smoothlabels,
                                               postriplet=postriplet, init_branches=init_branches,
                                               freeze_branches=freeze_branches, aux_losses=aux_losses, fMerge=fMerge,
                                               normbfmerge=normbfmerge, fActivation=fActivation, alpha=alpha,
                                               gaitset=gaitset, seed=seed)
        if initnet == "":
            model = build()
        else:
            model_base = UWYHSemiNet3Mods.loadnet(initnet)
            if model_base.get_layer("classprob").units != nclasses:
                print("Surgery needed: {} vs {}".format(model_base.get_layer("classprob").units, nclasses))
                model = _surgery(UWYHSemiNet3Mods, initnet, build,
                                 dict(nclasses=nclasses, loss_weights=loss_weights, dropout=dropout, margin=margin,
                                      optimizer=optimizer if optimizer is not None else optimizers.SGD(0.001, 0.9), use3D=use3D,
                                      postriplet=postriplet, fMerge=fMerge, fActivation=fActivation, gaitset=gaitset))
            else:
                model = model_base
            model = _freeze_like_reference(model, build, initnet, freeze_convs, freeze_all, UWYHSemiNet.get_weights_filename)
        print("Alright")
        return model
