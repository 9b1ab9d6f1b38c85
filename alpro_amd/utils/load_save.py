"""Checkpoint loading for the hot-path modules: same entry point and semantics as the reference's
`load_state_dict_with_pos_embed_resizing` (src/utils/load_save.py:73-134) and the two resize helpers it calls
(src/modeling/timesformer/helpers.py:355-378).

A released ALPRO / TimeSformer checkpoint carries `pos_embed (1, 1 + P, D)` and `time_embed (1, F, D)` for the grid it was
trained on; loading it into a model with another grid (other resolution / frame count) resamples both tables with
NEAREST-neighbour interpolation along the flattened patch axis / the frame axis, keeps the CLS slot, and then loads every
key whose shape matches (shape-mismatched keys and task heads are skipped, not errors), optionally renaming
`text_encoder.bert.*` -> `text_encoder.*` for the down-stream models.  Everything here is host-side, checkpoint-time work.
"""
import logging

import torch

LOGGER = logging.getLogger(__name__)


def nearest_index(n_in, n_out, device=None):
    """Source index of each of n_out samples under torch's legacy 'nearest' resampling of n_in samples:
    floor(i * n_in / n_out) in fp32, clamped -- the rule F.interpolate(mode='nearest') applies on every axis."""
    scale = torch.tensor(float(n_in) / float(n_out), dtype=torch.float32)
    idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * scale).to(torch.int64)
    return idx.clamp_(max=n_in - 1).to(device) if device is not None else idx.clamp_(max=n_in - 1)


def resize_spatial_embedding(state_dict, key, num_patches):
    """pos_embed (1, 1 + P, D) -> (1, 1 + num_patches, D): CLS slot kept, patch slots resampled along the flattened axis."""
    pos = state_dict[key]
    LOGGER.info("Resizing spatial position embedding from %d to %d", pos.size(1), num_patches + 1)
    idx = nearest_index(pos.size(1) - 1, num_patches, pos.device)
    return torch.cat((pos[:, :1], pos[:, 1:].index_select(1, idx)), 1)


def resize_temporal_embedding(state_dict, key, num_frames):
    """time_embed (1, F, D) -> (1, num_frames, D)."""
    te = state_dict[key]
    LOGGER.info("Resizing temporal position embedding from %d to %d", te.size(1), num_frames)
    return te.index_select(1, nearest_index(te.size(1), num_frames, te.device))


def load_state_dict_with_pos_embed_resizing(model, loaded_state_dict_or_path, num_patches, num_frames,
                                            spatial_embed_key='visual_encoder.model.pos_embed',
                                            temporal_embed_key='visual_encoder.model.time_embed',
                                            strict=False, remove_text_encoder_prefix=False):
    """In place on `model`.  See the module docstring; argument names and defaults follow the reference."""
    if isinstance(loaded_state_dict_or_path, str):
        loaded = torch.load(loaded_state_dict_or_path, map_location="cpu")
    else:
        loaded = loaded_state_dict_or_path
    loaded = dict(loaded)
    if remove_text_encoder_prefix:
        for key in list(loaded):
            if 'text_encoder.bert' in key:
                loaded[key.replace('text_encoder.bert', 'text_encoder')] = loaded.pop(key)
    if num_patches + 1 != loaded[spatial_embed_key].size(1):
        loaded[spatial_embed_key] = resize_spatial_embedding(loaded, spatial_embed_key, num_patches)
    if temporal_embed_key in loaded and num_frames != loaded[temporal_embed_key].size(1):
        loaded[temporal_embed_key] = resize_temporal_embedding(loaded, temporal_embed_key, num_frames)
    own = model.state_dict()
    toload, mismatched = {}, []
    for k, v in own.items():
        if k in loaded:
            if v.shape != loaded[k].shape:
                mismatched.append(k)
            else:
                toload[k] = loaded[k]
    LOGGER.info("Keys in loaded but not in model: %s", sorted(set(loaded) - set(own)))
    LOGGER.info("Keys in model but not in loaded: %s", sorted(set(own) - set(loaded)))
    LOGGER.info("Keys in model and loaded, but shape mismatched: %s", sorted(mismatched))
    model.load_state_dict(toload, strict=strict)
    return mismatched


# ---- savers / restorers the reference's drivers construct (src/utils/load_save.py:45-70, 202-347) --------------------------------
# Same constructor arguments, file names (`{prefix}_step_{step}.pt`, `{prefix}_step_{step}_train_state.pt`, `restore.pt`,
# `restore_backup.pt`) and dictionary keys, so checkpoints are interchangeable with the reference's in both directions.  Optimizer
# state goes through `optimizer.state_dict()` / `load_state_dict()` -- alpro_amd.optim.FlatAdamW implements both on its flat buffers.
def _map_tensors(state, fn):
    if torch.is_tensor(state):
        return fn(state)
    if isinstance(state, dict):
        return {k: _map_tensors(v, fn) for k, v in state.items()}
    if isinstance(state, (list, tuple)):
        return type(state)(_map_tensors(v, fn) for v in state)
    return state


def _to_cpu(state, narrow=True):
    """Host copy for a checkpoint.  Model tensors are narrowed fp32 -> fp16 like the reference's restorer files
    (load_save.py:181-196).  Optimizer state is NOT (narrow=False): Adam's second moments sit around 1e-10..1e-6, below fp16's
    range, and come back as zeros -- the reference narrows them too (a restore there restarts with v == 0); the files stay
    loadable by either side because both widen whatever they find."""
    return _map_tensors(state, lambda t: t.detach().cpu().half() if (narrow and t.dtype == torch.float32) else t.detach().cpu())


def _to_device(state, device=None):
    """Back onto the current device; fp16 tensors widen to fp32 (load_save.py:162-178)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    return _map_tensors(state, lambda t: t.to(device).float() if t.dtype == torch.float16 else t.to(device))


def _with_retries(what, fn, trials):
    """Blob stores fail now and then: the reference retries every save / restore up to 10 times and moves on; so do we,
    but the last error is logged instead of being dropped."""
    err = None
    for n in range(trials):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 -- any failure is retried, like the reference
            err = e
            LOGGER.warning("%s: trial %d failed: %r", what, n, e)
    LOGGER.error("%s: giving up after %d trials: %r", what, trials, err)
    return None


def save_training_meta(args):
    """What every driver calls once on rank 0 before training (load_save.py:19-42; run_pretrain_sparse.py:443, run_video_retrieval.py:
    363): `{output_dir}/log` and `/ckpt` exist afterwards, `log/args.json` holds the run arguments, `log/model_config.json` a copy of
    the model config file, and `code.zip` a snapshot of the code base (directories named __pycache__ / output / data / ext or containing
    'results', and *.pyc / *.ipynb / *.swap / *.pt files, are left out).  `args` is an EasyDict or any mapping / attribute bag with
    `output_dir` and `model_config`."""
    import json
    import os
    import zipfile
    get = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
    out = get("output_dir")
    os.makedirs(os.path.join(out, "log"), exist_ok=True)
    os.makedirs(os.path.join(out, "ckpt"), exist_ok=True)
    plain = dict(args) if isinstance(args, dict) else dict(vars(args))
    with open(os.path.join(out, "log", "args.json"), "w") as f:
        json.dump(plain, f, indent=4, sort_keys=True, default=str)
    with open(get("model_config")) as f:
        model_config = json.load(f)
    with open(os.path.join(out, "log", "model_config.json"), "w") as f:
        json.dump(model_config, f, indent=4, sort_keys=True)
    code_dir = os.environ.get("ALPRO_CODE_DIR") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    zip_path = os.path.join(out, "code.zip")
    LOGGER.info("Saving code from %s to %s...", code_dir, zip_path)
    skip_dirs, skip_ext = {"__pycache__", "output", "data", "ext", ".git", "gpurun_out"}, {".pyc", ".ipynb", ".swap", ".pt", ".so", ".o"}
    root = os.path.abspath(code_dir)
    with zipfile.ZipFile(zip_path, "w") as zf:
        for d, subdirs, files in os.walk(root):
            subdirs[:] = [x for x in subdirs if x not in skip_dirs and "results" not in x]
            for name in files:
                if os.path.splitext(name)[1] in skip_ext:
                    continue
                full = os.path.join(d, name)
                if os.path.realpath(full) == os.path.realpath(zip_path):
                    continue
                zf.write(full, os.path.join("code", os.path.relpath(full, root)))
    LOGGER.info("Saving code done.")


def compare_dict_difference(dict1, dict2, dict1_name="dict1", dict2_name="dict2", print_value_diff=True, verbose=False):
    """(values that differ on shared keys as {key: [(name1, v1), (name2, v2)]}, keys present in only one of the two) --
    load_save.py:138-176; the restorers use it to report changed run arguments."""
    import json
    k1, k2 = set(dict1), set(dict2)
    shared = k1 & k2
    only1, only2 = k1 - shared, k2 - shared
    value_diff = {k: [(dict1_name, dict1[k]), (dict2_name, dict2[k])] for k in shared if dict1[k] != dict2[k]}
    if verbose:
        LOGGER.info("keys in %s but not in %s: total %d, %s", dict1_name, dict2_name, len(only1), sorted(only1))
        LOGGER.info("keys in %s but not in %s: total %d, %s", dict2_name, dict1_name, len(only2), sorted(only2))
        if print_value_diff:
            LOGGER.info("%s", json.dumps(value_diff, indent=4, default=str))
    return value_diff, list(only1) + list(only2)


class ModelSaver:
    """`{output_dir}/{prefix}_step_{step}.pt` = model.state_dict() on the CPU; with an optimizer also
    `..._train_state.pt` = {'step', 'optimizer'} (load_save.py:45-70)."""

    def __init__(self, output_dir):
        self.output_dir = output_dir
        self.max_save_load_trial = 10

    def save(self, step, model, optimizer=None, prefix="model"):
        import os

        def once():
            sd = {k: v.cpu() if torch.is_tensor(v) else v for k, v in model.state_dict().items()}
            torch.save(sd, os.path.join(self.output_dir, "%s_step_%s.pt" % (prefix, step)))
            if optimizer is not None:
                osd = _map_tensors(optimizer.state_dict(), lambda t: t.detach().cpu())
                torch.save({"step": step, "optimizer": osd}, os.path.join(self.output_dir, "%s_step_%s_train_state.pt" % (prefix, step)))
            return True
        return _with_retries("ModelSaver.save", once, self.max_save_load_trial)


class _RestorerBase:
    def _init_paths(self, opts):
        self.save_path = "%s/restore.pt" % opts.output_dir
        self.backup_path = "%s/restore_backup.pt" % opts.output_dir
        self.amp = getattr(opts, "fp16", 0)
        self.max_save_load_trial = 10

    def _resume_or_start(self):
        import os
        if os.path.exists(self.save_path) or os.path.exists(self.backup_path):
            LOGGER.info("found previous checkpoint. try to resume...")
            self.global_step = None
            _with_retries(type(self).__name__ + ".restore", self.restore, self.max_save_load_trial)
            if self.global_step is None:
                raise RuntimeError("%s: restore.pt exists but could not be restored (see the log)" % type(self).__name__)
        else:
            self.global_step = 0

    def _read(self):
        try:
            return torch.load(self.save_path, map_location="cpu")
        except Exception:  # noqa: BLE001 -- a torn restore.pt falls back to the previous one
            return torch.load(self.backup_path, map_location="cpu")

    def _amp_active(self):
        """Loss-scaler state belongs in the checkpoint whenever a scaler exists -- `fp16: 1` in the config like the reference
        (load_save.py:238,262), OR fp16 operands chosen by the launcher (`--dtype fp16`, the default; every release config says fp16: 0):
        without it a resumed run restarts at 2^16 and skips steps until the scale has re-adapted."""
        from alpro_amd import amp
        return bool(self.amp) or amp.needs_loss_scaling() or bool(amp._SCALERS)

    def _write(self, checkpoint):
        """Two generations on disk, and a failed / interrupted save never costs one of them: the new checkpoint is written to
        `restore.pt.tmp` and fsync'ed FIRST; only then does the current restore.pt become the backup and the new file take its
        name (both os.replace, atomic on POSIX).  The reference renames before it saves (load_save.py:255-258, 316-319), so a torn
        save followed by its own retry renames the torn file over the only good backup."""
        import os
        if self._amp_active():
            from alpro_amd import amp
            checkpoint["amp_state_dict"] = amp.state_dict()
        tmp = self.save_path + ".tmp"
        with open(tmp, "wb") as f:
            torch.save(checkpoint, f)
            f.flush()
            os.fsync(f.fileno())
        if os.path.exists(self.save_path):
            os.replace(self.save_path, self.backup_path)
        os.replace(tmp, self.save_path)

    def step(self):
        self.global_step += 1
        if self.global_step % self.save_steps == 0:
            _with_retries(type(self).__name__ + ".save", self.save, self.max_save_load_trial)


class TrainingRestorer(_RestorerBase):
    """Generic restorer over named state holders (load_save.py:202-277): restore.pt = {'global_step', <name>: state_dict, ...}."""

    def __init__(self, opts, **ckpt_dict):
        self._init_paths(opts)
        self.ckpt_dict = ckpt_dict
        self.save_steps = opts.save_steps
        self._resume_or_start()

    def save(self):
        ck = {"global_step": self.global_step}
        for k, holder in self.ckpt_dict.items():
            ck[k] = _to_cpu(holder.state_dict(), narrow=not hasattr(holder, "param_groups"))
        self._write(ck)

    def restore(self):
        ck = self._read()
        for k, holder in self.ckpt_dict.items():
            holder.load_state_dict(_to_device(ck[k]))
        if self._amp_active() and ck.get("amp_state_dict") is not None:
            from alpro_amd import amp
            amp.load_state_dict(ck["amp_state_dict"])
        self.global_step = ck["global_step"]
        LOGGER.info("resume training from step %d", self.global_step)


class E2E_TrainingRestorer(_RestorerBase):
    """The pretraining / finetuning drivers' restorer (load_save.py:280-347): restore.pt = {'global_step', 'model_state_dict',
    'optim_state_dict'[, 'amp_state_dict']}, written every save_steps_ratio * num_train_steps steps."""

    def __init__(self, opts, model, optimizer):
        import json
        import os
        args_json = "%s/log/args.json" % opts.output_dir
        if os.path.exists(args_json):
            with open(os.path.join(opts.output_dir, "log", "restore_args.json"), "w") as w:
                json.dump(dict(vars(opts)) if not isinstance(opts, dict) else dict(opts), w, indent=4, default=str)
        self._init_paths(opts)
        self.model, self.optimizer = model, optimizer
        self.save_steps = max(1, int(opts.save_steps_ratio * opts.num_train_steps))
        self._resume_or_start()

    def save(self):
        self._write({"global_step": self.global_step, "model_state_dict": _to_cpu(self.model.state_dict()),
                     "optim_state_dict": _to_cpu(self.optimizer.state_dict(), narrow=False)})

    def restore(self, opts=None):
        ck = self._read()
        self.model.load_state_dict(_to_device(ck["model_state_dict"]))
        self.optimizer.load_state_dict(_to_device(ck["optim_state_dict"]))
        if self._amp_active() and ck.get("amp_state_dict") is not None:
            from alpro_amd import amp
            amp.load_state_dict(ck["amp_state_dict"])
        self.global_step = ck["global_step"]
        LOGGER.info("resume training from step %d", self.global_step)
