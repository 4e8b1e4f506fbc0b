"""3dunetcnn_amd: MI355X-native (gfx950) 3D U-Net hot path -- drop-in for ellisdg/3DUnetCNN's model/loss/optimizer step.

Import with importlib (the directory name starts with a digit): `importlib.import_module("3dunetcnn_amd")`.
"""
from . import _lib  # noqa: F401
from ._lib import LIB_PATH, load_library  # noqa: F401

__version__ = "0.1.0"
