"""Import alias: the product package directory is named after the reference repository
(`reconstruction-network-for-video-captioning_amd`), which is not a valid Python identifier.
`import recnet_amd` gives the same module object."""
import importlib
import sys

_pkg = importlib.import_module("reconstruction-network-for-video-captioning_amd")
sys.modules[__name__] = _pkg
