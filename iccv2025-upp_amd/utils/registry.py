"""Name -> class registry with the mmcv-style surface the reference uses
(reference utils/registry.py: Registry:6-243, build_from_cfg:246-287)."""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    """cfg['NAME'] selects the class; it is constructed as cls(cfg)."""
    if not isinstance(cfg, dict):
        raise TypeError(f'a model config is a dict (EasyDict), not {type(cfg).__name__}')
    if 'NAME' not in cfg and (default_args is None or 'NAME' not in default_args):
        raise KeyError(f'no "NAME" entry names the class to build: cfg = {cfg}, default_args = {default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'build_from_cfg looks classes up in a Registry, not in a {type(registry).__name__}')
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f'default_args: a dict of extra config entries or None, not {type(default_args).__name__}')
    if default_args is not None:
        for k, v in default_args.items():
            cfg[k] = v
    obj_type = cfg.get('NAME')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'the {registry.name} registry holds no class called {obj_type}')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'"NAME" is a registered name or a class, not {type(obj_type).__name__}')
    try:
        return obj_cls(cfg)
    except Exception as e:  # the plain exception does not name the class
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self._scope = scope
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
        self.parent = parent

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={self._module_dict})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'only classes can be registered, not {type(module_class).__name__}')
        names = [module_name or module_class.__name__] if not isinstance(module_name, (list, tuple)) else module_name
        for name in names:
            if not force and name in self._module_dict:
                raise KeyError(f'the {self.name} registry already has a class called {name} (force=True replaces it)')
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force is True or False, not {type(force).__name__}')
        if not (name is None or isinstance(name, str) or (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name: None (the class name), one string or several, not {type(name).__name__}')
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _register(cls):
            self._register_module(cls, name, force)
            return cls

        return _register
