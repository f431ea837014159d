"""YAML -> attribute dict with ``_base_`` includes (reference: utils/config.py:18-58).
easydict is not a dependency: EasyDict below is a minimal equivalent."""
import os
import shutil
import yaml


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__

    def __delattr__(self, k):
        del self[k]


_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resolve(path):
    """``_base_`` paths are written relative to the repo root in the reference ('cfgs/...');
    accept that, an absolute path, or a path relative to the act_amd package."""
    for cand in (path, os.path.join(_PKG_ROOT, path), os.path.join(os.path.dirname(_PKG_ROOT), path)):
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(path)


def merge_new_config(config, new_config):
    for key, val in new_config.items():
        if not isinstance(val, dict):
            if key == "_base_":
                with open(_resolve(new_config["_base_"]), "r") as f:
                    val = yaml.safe_load(f)
                config[key] = EasyDict()
                merge_new_config(config[key], val)
            else:
                config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file):
    config = EasyDict()
    with open(_resolve(cfg_file), "r") as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config=config, new_config=new_config)
    return config


def get_config(args, logger=None):
    from .logger import print_log
    if args.resume:
        cfg_path = os.path.join(args.experiment_path, "config.yaml")
        if not os.path.exists(cfg_path):
            print_log("Failed to resume", logger=logger)
            raise FileNotFoundError()
        print_log(f"Resume yaml from {cfg_path}", logger=logger)
        args.config = cfg_path
    config = cfg_from_yaml_file(args.config)
    if not args.resume and args.local_rank == 0:
        save_experiment_config(args, config, logger)
    apply_fewshot_args(args, config)
    return config


def apply_fewshot_args(args, config):
    """--way / --shot / --fold select the few-shot split: copied into the train and val dataset sections as the reference's entry script
    does right after get_config (main.py:72-78).  Idempotent; a no-op when --shot is -1 / absent or the recipe has no dataset section."""
    shot = getattr(args, "shot", -1)
    if shot is None or shot == -1 or "dataset" not in config:
        return config
    for split in ("train", "val"):
        sec = config.dataset.get(split)
        if sec is None:
            continue
        if "others" not in sec:
            sec.others = EasyDict()
        sec.others.shot, sec.others.way, sec.others.fold = shot, getattr(args, "way", -1), getattr(args, "fold", -1)
    return config


def save_experiment_config(args, config, logger=None):
    from .logger import print_log
    config_path = os.path.join(args.experiment_path, "config.yaml")
    shutil.copyfile(_resolve(args.config), config_path)
    print_log(f"Copy the Config file from {args.config} to {config_path}", logger=logger)
