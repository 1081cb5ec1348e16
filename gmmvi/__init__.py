"""Drop-in alias: ``import gmmvi...`` resolves to the MI355X implementation in ``gmmvi_amd`` so that the reference's
examples (``from gmmvi.gmmvi_runner import GmmviRunner``, ``from gmmvi.configs import ...``) run unchanged."""
import importlib
import importlib.abc
import importlib.util
import sys


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == "gmmvi" or not fullname.startswith("gmmvi."):
            return None
        real = "gmmvi_amd." + fullname[len("gmmvi."):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        real = "gmmvi_amd." + spec.name[len("gmmvi."):]
        return importlib.import_module(real)

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
from gmmvi_amd import __version__  # noqa: E402,F401
