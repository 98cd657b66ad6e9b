"""Hydra-free configuration shim.

The reference drives everything through Hydra 1.1 (``conf/config.yaml`` + config groups + ``expts/*.txt`` override
lists, train_net.py:17-35, launch.py:169-182) and builds objects with ``hydra.utils.instantiate``.  hydra/omegaconf
are not installed on the MI355X image, so this module implements the subset those files use:

  * ``compose(conf_dir, overrides)``: ``defaults`` lists (``group: option``, ``group@package: option``, ``override``
    entries for hydra's own groups are ignored), ``# @package _group_`` placement, ``key=value`` / ``+key=value`` /
    ``group/option=name`` overrides, ``${a.b}`` interpolation and the ``minus`` / ``times_int`` resolvers
    (train_net.py:17-19), ``${cwd}`` / ``${hydra:runtime.cwd}``.
  * ``instantiate(node, *args, **kwargs)``: ``_target_`` construction with keyword merging, recursing into nested
    ``_target_`` nodes unless ``_recursive_=False``.  Reference ``_target_`` strings (``models.*``, ``func.*``,
    ``loss_fn.*``, ``common.*``) resolve to this package's mirrors, so the reference's YAML works unchanged.
When real Hydra is importable the reference's own entry point can be used instead; nothing here depends on it.
"""
import copy
import importlib
import math
import os
import re

import yaml

_MIRRORS = ('models.', 'func.', 'loss_fn.', 'common.')
_SUBSTITUTES = {'torch.nn.Linear': 'avt_amd.models.classifiers.HipLinear'}


class Cfg(dict):
    """dict with attribute access (stands in for an OmegaConf DictConfig)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_cfg(obj):
    if isinstance(obj, dict):
        return Cfg({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_cfg(v) for v in obj]
    return obj


def locate(path: str, substitute=True):
    if substitute and path in _SUBSTITUTES:
        path = _SUBSTITUTES[path]
    if path.startswith(_MIRRORS):
        path = 'avt_amd.' + path
    mod, _, attr = path.rpartition('.')
    return getattr(importlib.import_module(mod), attr)


def instantiate(node, *args, **kwargs):
    recursive = kwargs.pop('_recursive_', True)
    if node is None:
        return None
    conf = {k: v for k, v in dict(node).items() if k not in ('_target_', '_recursive_')}
    if recursive:
        conf = {k: (instantiate(v) if isinstance(v, dict) and '_target_' in v else v) for k, v in conf.items()}
    conf.update(kwargs)
    return locate(node['_target_'])(*args, **conf)


call = instantiate


# ---- composition -----------------------------------------------------------------------------------------------------
def _load_yaml(path):
    with open(path) as f:
        text = f.read()
    pkg = None
    m = re.search(r'#\s*@package\s+(\S+)', text)
    if m:
        pkg = m.group(1)
    return (yaml.safe_load(text) or {}), pkg


def _set(cfg, dotted, value, create=True):
    keys = dotted.split('.')
    cur = cfg
    for k in keys[:-1]:
        if k not in cur or not isinstance(cur[k], dict):
            if not create:
                raise KeyError(dotted)
            cur[k] = {}
        cur = cur[k]
    cur[keys[-1]] = value


def _get(cfg, dotted):
    cur = cfg
    for k in dotted.split('.'):
        cur = cur[k]
    return cur


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def _parse_value(text):
    text = text.strip()
    if '${' in text:
        return text                      # resolved later
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def read_overrides(path):
    """``expts/*.txt``: one override per line, ``#`` comments (launch.py:169-182)."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.split('#', 1)[0].strip() if not line.strip().startswith('#') else ''
            if line:
                out.append(line)
    return out


_RESOLVERS = {
    'minus': lambda a, b: _num(a) - _num(b),
    'times_int': lambda a, b: int(_num(a) * _num(b)),
}


def _num(x):
    x = yaml.safe_load(str(x))
    return x


def _resolve_str(s, root, depth=0, path=()):
    """``path`` = key path of the node holding ``s`` (needed for OmegaConf-style relative keys ``${.a}`` / ``${..a}``)."""
    if depth > 20:
        raise ValueError(f'interpolation too deep: {s}')
    pat = re.compile(r'\$\{([^${}]+)\}')
    while True:
        m = pat.search(s) if isinstance(s, str) else None
        if not m:
            return s
        expr = m.group(1)
        if ':' in expr:
            name, argstr = expr.split(':', 1)
            if name == 'hydra':
                val = os.getcwd()
            else:
                args = [_resolve_str(a.strip(), root, depth + 1, path) for a in argstr.split(',')]
                val = _RESOLVERS[name](*args)
        elif expr == 'cwd':
            val = os.getcwd()
        else:
            if expr.startswith('.'):
                ndots = len(expr) - len(expr.lstrip('.'))
                base = list(path[:-1])
                base = base[:len(base) - (ndots - 1)] if ndots > 1 else base
                target = '.'.join(base + [expr.lstrip('.')])
            else:
                target = expr
            val = _resolve_str(_get(root, target), root, depth + 1, tuple(target.split('.')))
        if m.start() == 0 and m.end() == len(s):
            return val
        s = s[:m.start()] + str(val) + s[m.end():]


def _resolve(node, root, path=()):
    if isinstance(node, dict):
        return {k: _resolve(v, root, path + (k,)) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, path) for v in node]
    if isinstance(node, str) and '${' in node:
        v = _resolve_str(node, root, 0, path)
        if isinstance(v, str):
            try:
                parsed = yaml.safe_load(v)
                if not isinstance(parsed, str):
                    v = parsed
            except yaml.YAMLError:
                pass
        return v if isinstance(v, str) else _resolve(v, root, path)
    return node


def compose(conf_dir, overrides=(), config_name='config.yaml'):
    base, _ = _load_yaml(os.path.join(conf_dir, config_name))
    defaults = base.pop('defaults', [])
    base.pop('hydra', None)
    group_choice, order = {}, []
    for d in defaults:
        if isinstance(d, str):
            continue
        (k, v), = d.items()
        if k.startswith('override '):
            continue
        group, _, pkg = k.partition('@')
        if k not in group_choice:
            order.append(k)
        group_choice[k] = (group, pkg or None, v)
    sets = []
    for ov in overrides:
        key, _, val = ov.partition('=')
        plus = key.startswith('+')
        key = key.lstrip('+~')
        group = key.partition('@')[0]
        if os.path.isdir(os.path.join(conf_dir, group)) and ('/' in key or key in group_choice or '@' in key):
            if key not in group_choice:
                order.append(key)
            g, _, pkg = key.partition('@')
            group_choice[key] = (g, pkg or None, val.strip())
        elif key.startswith('hydra.') or key.startswith('hydra/'):
            continue
        else:
            sets.append((key, val, plus))
    cfg = {}
    for k in order:
        group, pkg, option = group_choice[k]
        if option in (None, 'null'):
            continue
        path = os.path.join(conf_dir, group, str(option) + '.yaml')
        if not os.path.exists(path):
            raise FileNotFoundError(f'config group option not found: {path}')
        data, hdr = _load_yaml(path)
        if pkg:
            where = pkg
        elif hdr in (None, '_group_'):
            where = group.replace('/', '.')
        elif hdr == '_global_':
            where = ''
        else:
            where = hdr
        if where:
            tmp = {}
            _set(tmp, where, data)
            _merge(cfg, tmp)
        else:
            _merge(cfg, data)
    _merge(cfg, base)          # primary config overrides group defaults (hydra's _self_-last convention)
    for key, val, plus in sets:
        _set(cfg, key, _parse_value(val), create=True)
    return to_cfg(_resolve(cfg, cfg))
