"""Config surface of ``run.py`` (reference: ``instageo/model/configs/*.yaml`` + Hydra ``key=value`` overrides).

Hydra/OmegaConf are not available offline, so the same keys are provided as plain dictionaries: ``DEFAULTS``
carries every key of the reference's ``config.yaml``; ``PRESETS[name]`` holds the values in which
``--config-name name`` differs.  Values are the reference's published recipe constants (data, not code).
"""
from __future__ import annotations

import copy
from typing import Any, Dict, List

DEFAULTS: Dict[str, Any] = {'root_dir': None,
 'valid_filepath': None,
 'train_filepath': None,
 'test_filepath': None,
 'checkpoint_path': None,
 'mode': 'train',
 'is_reg_task': False,
 'train': {'learning_rate': 0.0001,
           'num_epochs': 10,
           'batch_size': 8,
           'class_weights': [1, 1],
           'ignore_index': -100,
           'weight_decay': 0.01,
           'scheduler': False,
           'distillation': False,
           'teacher_ckpt_path': None},
 'model': {'model_name': 'prithvi_eo_tiny',
           'freeze_backbone': False,
           'load_pretrained_weights': True,
           'num_classes': 2,
           'use_log_scale': False,
           'plot_reg_results': False,
           'include_ee_metric': False,
           'weight_clip_range': None,
           'depth': -1},
 'dataloader': {'bands': [1, 2, 3, 8, 11, 12],
                'mean': [0.14245495, 0.13921481, 0.12434631, 0.31420089, 0.20743526, 0.12046503],
                'std': [0.04036231, 0.04186983, 0.05267646, 0.0822221, 0.06834774, 0.05294205],
                'img_size': 224,
                'temporal_dim': 1,
                'replace_label': [-1, 2],
                'reduce_to_zero': False,
                'no_data_value': -9999,
                'constant_multiplier': 1.0,
                'max_pixel_value': 10000,
                'num_workers': 1,
                'augmentations': {'hflip': {'use': True, 'p': 0.5},
                                  'vflip': {'use': True, 'p': 0.5},
                                  'rotate': {'use': True, 'p': 0.5, 'degrees': 10},
                                  'brightness': {'use': True,
                                                 'p': 0.5,
                                                 'brightness_range': [0.8, 1.2],
                                                 'contrast_range': [0.8, 1.2]},
                                  'blur': {'use': True, 'p': 0.5, 'kernel_size': 3, 'sigma_range': [0.1, 2.0]},
                                  'noise': {'use': True, 'p': 0.5, 'noise_std': 0.05}}},
 'test': {'img_size': 224, 'crop_size': 224, 'stride': 224, 'mask_cloud': False}}

PRESETS: Dict[str, Dict[str, Any]] = {'sen1floods11': {'train': {'batch_size': 16, 'class_weights': [1, 3], 'ignore_index': -1},
                  'model': {'model_name': 'prithvi_eo_v1_100'},
                  'dataloader': {'replace_label': None,
                                 'augmentations': {'rotate': {'use': False},
                                                   'brightness': {'use': False},
                                                   'blur': {'use': False},
                                                   'noise': {'use': False}}},
                  'test': {'img_size': 512}},
 'multitemporal_crop_classification': {'train': {'class_weights': [0.386375,
                                                                   0.661126,
                                                                   0.548184,
                                                                   0.640482,
                                                                   0.876862,
                                                                   0.925186,
                                                                   3.249462,
                                                                   1.542289,
                                                                   2.175141,
                                                                   2.272419,
                                                                   3.062762,
                                                                   3.626097,
                                                                   1.198702],
                                                 'ignore_index': -1},
                                       'model': {'model_name': 'prithvi_eo_v1_100', 'num_classes': 13},
                                       'dataloader': {'bands': [0,
                                                                1,
                                                                2,
                                                                3,
                                                                4,
                                                                5,
                                                                6,
                                                                7,
                                                                8,
                                                                9,
                                                                10,
                                                                11,
                                                                12,
                                                                13,
                                                                14,
                                                                15,
                                                                16,
                                                                17],
                                                      'mean': [494.905781,
                                                               815.239594,
                                                               924.335066,
                                                               2968.881459,
                                                               2634.621962,
                                                               1739.579917],
                                                      'std': [284.925432,
                                                              357.84876,
                                                              575.566823,
                                                              896.601013,
                                                              951.900334,
                                                              921.407808],
                                                      'temporal_dim': 3,
                                                      'replace_label': None,
                                                      'reduce_to_zero': True,
                                                      'no_data_value': None}},
 'locust': {'train': {'num_epochs': 20, 'ignore_index': -1, 'weight_decay': 0.1},
            'model': {'model_name': 'prithvi_eo_v1_100'},
            'dataloader': {'bands': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17],
                           'mean': [623.2724609375,
                                    1247.657958984375,
                                    1772.24169921875,
                                    2371.256103515625,
                                    2862.867431640625,
                                    2357.759765625],
                           'std': [2182.050048828125,
                                   2248.420654296875,
                                   2302.53515625,
                                   2372.204345703125,
                                   2398.52685546875,
                                   2292.96435546875],
                           'temporal_dim': 3,
                           'replace_label': [-9999, -1],
                           'augmentations': {'rotate': {'use': False},
                                             'brightness': {'use': False},
                                             'blur': {'use': False},
                                             'noise': {'use': False}}}}}
PRESETS["config"] = {}


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _parse_value(text: str) -> Any:
    import yaml

    try:
        return yaml.safe_load(text)
    except Exception:
        return text


def load_config(config_name: str = "config", overrides: List[str] = (), config_path: str = None) -> Dict[str, Any]:
    """``--config-name`` preset (or a YAML file under ``--config-path``) + ``a.b.c=value`` / ``+key=value`` overrides."""
    cfg = copy.deepcopy(DEFAULTS)
    if config_path:
        import os

        import yaml

        path = os.path.join(config_path, config_name if config_name.endswith((".yaml", ".yml")) else config_name + ".yaml")
        _merge(cfg, yaml.safe_load(open(path)) or {})
    elif config_name not in PRESETS:
        raise KeyError(f"unknown config {config_name!r}; available: {sorted(PRESETS)}")
    else:
        _merge(cfg, copy.deepcopy(PRESETS[config_name]))
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not key=value")
        key, val = ov.split("=", 1)
        add = key.startswith("+")
        key = key.lstrip("+")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node:
                if not add:
                    raise KeyError(f"unknown config key {key!r} (use +{key}=... to add)")
                node[p] = {}
            node = node[p]
        if parts[-1] not in node and not add:
            raise KeyError(f"unknown config key {key!r} (use +{key}=... to add)")
        node[parts[-1]] = _parse_value(val)
    return cfg


def check_required_flags(required: List[str], cfg: Dict[str, Any]) -> None:
    """pipeline_utils.py:44-55: a flag counts as missing when it is None or the *string* "None"."""
    for flag in required:
        if cfg.get(flag) is None or cfg.get(flag) == "None":
            raise RuntimeError(f"Flag --{flag} is required.")
