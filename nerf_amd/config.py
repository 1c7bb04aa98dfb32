"""YAML configuration with the reference's keys and CLI (misc.py:10-35) without OmegaConf:
``--configs a.yaml b.yaml`` merged left to right, ``--params key.sub=value ...`` dot-list overrides.
Attribute access (cfg.trainer.n_rays) works like the reference's OmegaConf tree."""
import argparse
import copy

import yaml


class Cfg(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def merge(a, b):
    """Recursive dict merge, b wins (OmegaConf.merge semantics for plain trees)."""
    out = copy.deepcopy(a)
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def from_dotlist(items):
    out = {}
    for it in items:
        key, _, val = it.partition('=')
        node = out
        parts = key.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return out


def _load(path):
    with open(path) as f:
        tree = yaml.safe_load(f) or {}
    return _fix_numbers(tree)


def _fix_numbers(x):
    """PyYAML reads '1e-5' as a string (no dot); the reference's configs use that spelling."""
    if isinstance(x, dict):
        return {k: _fix_numbers(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_fix_numbers(v) for v in x]
    if isinstance(x, str):
        try:
            return float(x) if any(c in x for c in 'eE.') and x.replace('-', '').replace('+', '').replace('.', '').replace('e', '').replace('E', '').isdigit() else x
        except ValueError:
            return x
    return x


def configurator(config_base=None, argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--configs', type=str, nargs='*')
    parser.add_argument('--params', type=str, nargs='*')
    args = parser.parse_args(argv)
    cfg = {}
    if args.configs is None and config_base is not None:
        cfg = merge(cfg, _load(config_base))
    else:
        for c in args.configs or []:
            cfg = merge(cfg, _load(c))
    if args.params is not None:
        cfg = merge(cfg, _fix_numbers(from_dotlist(args.params)))
    return _wrap(cfg)
