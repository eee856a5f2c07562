"""Import alias: the product package lives in ``seed-x_amd/`` (a directory name Python cannot import
directly because of the hyphen). ``import seedx_amd`` executes that package under this name."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "seed-x_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"), globals())
