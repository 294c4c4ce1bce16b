"""Sampling engine: checkpoint -> generator -> images  (SURVEY §8f N3).

Mirrors the BigGAN branch of the reference's `inference/utils.py`:
  sample                 inference/utils.py:176-269   one conditioning draw -> G(z, y, feats) under no_grad
  load_model_inference   inference/utils.py:272-403   pick best checkpoint by FID, adopt its config, build G, load
                                                       G or G_ema weights, eval mode
Checkpoints are the reference's own files (`{G,G_ema,D,G_optim,D_optim,state_dict}[_suffix].pth`,
BigGAN_PyTorch/utils.py:1116-1167): a checkpoint written by the reference loads here and vice versa
(tests/test_checkpoint.py runs both directions against files written by the reference).

The StyleGAN2 branch (pickled network, inference/utils.py:395-403) is not part of this path and raises.
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
    if backbone != "biggan":
        raise NotImplementedError("only the BigGAN backbone is served by this engine (backbone=%r)" % (backbone,))
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
        gen_samples = generator(z_, y_, feats_)
    return gen_samples, y_, feats_


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
    if config.get("model_backbone", "biggan") != "biggan":
        raise NotImplementedError("only model_backbone='biggan' is served by this engine")
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
    return generator, config
