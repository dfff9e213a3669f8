"""Import alias: the package directory is ``knn-for-homology_amd/`` (the name the
project layout prescribes); a hyphen cannot appear in an import statement, so this
module makes ``import knn_for_homology_amd`` resolve to that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "knn-for-homology_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
