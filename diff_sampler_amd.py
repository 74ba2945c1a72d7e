"""Import shim: the package directory is `diff-sampler_amd/` (a hyphen is not a
valid Python identifier), so `import diff_sampler_amd` resolves here and this
module turns itself into a package rooted at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "diff-sampler_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
