"""Importable alias of the ``hand-gesture-recognition_amd/`` package directory.

The package directory carries the reference repository's name, which is not a valid Python
identifier; this alias executes its ``__init__`` with ``__path__`` pointing there, so
``import hand_gesture_recognition_amd.multitasknet`` resolves to
``hand-gesture-recognition_amd/multitasknet.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "hand-gesture-recognition_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
