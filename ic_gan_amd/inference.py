"""Sampling engine: checkpoint -> generator -> images  (SURVEY §8f N3).

Mirrors the BigGAN branch of the reference's `inference/utils.py`:
  sample                 inference/utils.py:176-269   one conditioning draw -> G(z, y, feats) under no_grad
  load_model_inference   inference/utils.py:272-403   pick best checkpoint by FID, adopt its config, build G, load
                                                       G or G_ema weights, eval mode
Checkpoints are the reference's own files (`{G,G_ema,D,G_optim,D_optim,state_dict}[_suffix].pth`,
BigGAN_PyTorch/utils.py:1116-1167): a checkpoint written by the reference loads here and vice versa
(tests/test_checkpoint.py runs both directions against files written by the reference).

The StyleGAN2 branch (inference/utils.py:395-403) reads the pickled `best-network-snapshot.pkl` through
ic_gan_amd.stylegan2.legacy.load_network_pkl (no execution of the source text embedded in the pickle) and returns G_ema.
"""
import torch

from . import utils as hot_utils

# keys of the caller's config that win over the checkpointed config (inference/utils.py:335-362)
_CALLER_KEYS = frozenset([
    "base_root", "data_root", "load_weights", "batch_size", "num_workers", "weights_root", "logs_root", "samples_root",
    "eval_reference_set", "eval_instance_set", "which_dataset", "seed", "eval_prdc", "use_balanced_sampler",
    "custom_distrib", "longtail_temperature", "longtail_gen", "num_inception_images", "sample_num_npz", "load_in_mem",
    "split", "z_var", "kmeans_subsampled", "filter_hd", "n_subsampled_data", "feature_augmentation",
])


def sample(generator, sample_conditioning_func, config, class_cond=True, instance_cond=False, device="cuda",
           backbone="biggan", truncation_value=1.0):
    """One batch of generated images; returns (gen_samples, y_, feats_) like inference/utils.py:176-269.

    `sample_conditioning_func()` yields z | (z, y) | (z, feats) | (z, y, feats) in that order
    (data_utils/utils.py:877-901)."""
    if backbone not in ("biggan", "stylegan2"):
        raise NotImplementedError("backbone must be 'biggan' or 'stylegan2' (got %r)" % (backbone,))
    if config.get("parallel", False):
        raise NotImplementedError("nn.DataParallel sampling is not supported; run one process per GPU")
    cond = sample_conditioning_func()
    with torch.no_grad():
        y_ = feats_ = None
        if class_cond and instance_cond:
            z_, y_, feats_ = cond
        elif class_cond:
            z_, y_ = cond
        elif instance_cond:
            z_, feats_ = cond
        else:
            z_ = cond
        if y_ is not None:
            y_ = y_.long().to(device, non_blocking=True)
        if feats_ is not None:
            feats_ = feats_.to(device, non_blocking=True)
        z_ = z_.to(device, non_blocking=True).as_subclass(torch.Tensor)
        if backbone == "stylegan2":          # inference/utils.py:243-262: one-hot / empty class vector, truncation, constant noise
            n = z_.shape[0]
            c = torch.empty([n, generator.c_dim], device=device) if y_ is None else torch.eye(config["n_classes"], device=device)[y_]
            h = torch.empty([n, generator.h_dim], device=device) if feats_ is None else feats_
            gen_samples = generator(z=z_, c=c, feats=h, truncation_psi=truncation_value, noise_mode="const")
            return gen_samples, c, h
        gen_samples = generator(z_, y_, feats_)
    return gen_samples, y_, feats_


class GraphedGenerator:
    """The eval-mode generator forward captured once in a HIP graph and replayed per call.

    At small batch sizes the forward is launch-bound (≈400 kernel launches, a few ms of host time for 1-2 ms of GPU work);
    the eval forward is static (stored BN statistics, no spectral-norm update, no autograd), so the whole launch sequence is
    recorded on a capture stream with fixed input/output buffers and replayed with one host call.  `sample(...)` accepts a
    GraphedGenerator in place of the module.  Changing batch size, conditioning kind or train/eval mode needs a new capture.
    """

    def __init__(self, generator, batch_size, class_cond=True, instance_cond=False, device="cuda", feature_dim=2048,
                 static_weights=False):
        """static_weights=False: the spectral-norm passes are part of the graph, replays always use the module's current
        weights.  static_weights=True: W/sigma of every layer is computed once before capture and baked into the graph
        (fewer, smaller replays — the right choice for serving a fixed checkpoint); call `refresh()` after loading weights."""
        if generator.training:
            raise RuntimeError("GraphedGenerator captures the eval-mode forward: call generator.eval() first")
        self._args = (generator, batch_size, class_cond, instance_cond, device, feature_dim)
        self.static_weights = static_weights
        self.refresh()

    def refresh(self):
        from . import layers
        was, layers.SN_EVAL_CACHE = layers.SN_EVAL_CACHE, bool(self.static_weights)
        # the per-layer opt-in flags of the CALLER's generator are restored after the capture: the graph keeps the W/sigma
        # buffers it baked in alive through self._pinned, while eval forwards of the module OUTSIDE the graph go back to
        # whatever caching policy its owner chose (a module that is written through `.data` must not serve a stale cache)
        sn_layers = [m for m in self._args[0].modules() if isinstance(m, layers.SN)]
        flags = [getattr(m, "_sn_cache_ok", False) for m in sn_layers]
        if self.static_weights:
            layers.enable_sn_eval_cache(self._args[0], True)      # fixed checkpoint: W/sigma computed once, baked in
        try:
            self._capture(*self._args)
        finally:
            layers.SN_EVAL_CACHE = was
            for m, f in zip(sn_layers, flags):
                m._sn_cache_ok = f
                if not f:
                    m._sn_eval = None                              # (the graph's copy lives on in self._pinned)

    def _capture(self, generator, batch_size, class_cond, instance_cond, device, feature_dim):
        self.generator, self.batch_size = generator, batch_size
        dev = torch.device(device)
        self._z = torch.zeros(batch_size, generator.dim_z, device=dev)
        self._y = torch.zeros(batch_size, dtype=torch.int64, device=dev) if class_cond else None
        self._f = torch.zeros(batch_size, feature_dim, device=dev) if instance_cond else None
        if self._f is not None:
            self._f[:, 0] = 1.0
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(2):                      # warm-up outside capture (lazy initialisation, allocator pools)
                generator(self._z, self._y, self._f)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._out = generator(self._z, self._y, self._f)
        # a graph holds raw pointers: the cached W/sigma buffers baked into it (static_weights) must outlive it even if
        # the modules drop them from their caches later
        self._pinned = [m._sn_eval for m in generator.modules() if getattr(m, "_sn_eval", None) is not None]

    def __call__(self, z, y=None, feats=None):
        self._z.copy_(z, non_blocking=True)
        if self._y is not None:
            self._y.copy_(y, non_blocking=True)
        if self._f is not None:
            self._f.copy_(feats, non_blocking=True)
        self._graph.replay()
        return self._out.clone()


def _best_checkpoint(config):
    """inference/utils.py:284-308: the `best0`/`best1` checkpoint with the lower recorded FID ('' if neither exists)."""
    root = "/".join([config["weights_root"], config["experiment_name"]])
    best, best_fid = "", 1e5
    for name in ("best0", "best1"):
        try:
            sd = torch.load("%s/%s.pth" % (root, hot_utils.join_strings("_", ["state_dict", name])), weights_only=False)
        except (OSError, RuntimeError):
            print("Checkpoint with name ", name, " not in folder.")
            continue
        if sd["best_FID"] < best_fid:
            best, best_fid = name, sd["best_FID"]
    return best


def load_model_inference(config, device="cuda"):
    """inference/utils.py:272-393 (BigGAN backbone): returns (generator, config) with `config` overwritten by the
    training-time configuration stored in the checkpoint, except for the caller-side keys."""
    if config.get("model_backbone", "biggan") == "stylegan2":
        # inference/utils.py:395-403: StyleGAN2 saves the entire network in a pickle
        import os
        from .stylegan2 import legacy
        network_pkl = os.path.join(config["base_root"], config["experiment_name"], "best-network-snapshot.pkl")
        print('Loading networks from "%s"...' % network_pkl)
        with open(network_pkl, "rb") as f:
            generator = legacy.load_network_pkl(f)["G_ema"].to(device)
        return generator, config
    if config.get("model_backbone", "biggan") != "biggan":
        raise NotImplementedError("model_backbone must be 'biggan' or 'stylegan2'")
    from . import BigGAN as model

    if not config.get("experiment_name"):
        raise ValueError("load_model_inference needs config['experiment_name'] (name_from_config is not part of this engine)")
    config["load_weights"] = _best_checkpoint(config)
    print("Final name selected is ", config["load_weights"])
    state_dict = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_IS": 0, "best_FID": 999999,
                  "config": config}
    hot_utils.load_weights(None, None, state_dict, config["weights_root"], config["experiment_name"],
                           config["load_weights"], None, strict=False, load_optim=False, eval=True)
    for item, value in state_dict["config"].items():
        if item in _CALLER_KEYS or (item == "experiment_name" and config["experiment_name"] != ""):
            continue
        config[item] = value
    config["feature_augmentation"] = config["hflips"] = config["DA"] = False      # no augmentation at test time

    generator = model.Generator(**config).to(device)
    use_ema = config.get("ema", False) and config.get("use_ema", False)
    hot_utils.load_weights(None if config.get("use_ema", False) else generator, None, state_dict, config["weights_root"],
                           config["experiment_name"], config["load_weights"], generator if use_ema else None,
                           strict=False, load_optim=False)
    if config.get("G_eval_mode", False):
        generator.eval()
    # a sampling generator's weights are frozen from here on: W/sigma of every layer is computed once and reused
    # (layers.invalidate_sn_cache(generator) after any manual weight edit through `.data`)
    from . import layers
    layers.enable_sn_eval_cache(generator, True)
    return generator, config
