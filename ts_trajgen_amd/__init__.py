"""Import alias: the package lives in ``ts-trajgen_amd/`` (a directory name Python cannot import
directly because of the hyphen); ``import ts_trajgen_amd`` resolves to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'ts-trajgen_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _os, _f
