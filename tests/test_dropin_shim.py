"""`bevrender_amd.dropin.install()` in front of the reference's own import block
(reference train.py:1-26) must hand the reference this repo's model and retrieval losses and leave the rest of its
`loss` package alone.  Runs only where the reference checkout exists (the build container); CPU only, in a child
process (the aliases must not leak into the test session).  Third-party packages the image lacks (wandb,
torchvision, timm, pytorch_metric_learning, yourdfpy) are stubbed as empty modules for the import block only."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_import_block_resolves_to_this_package():
    code = textwrap.dedent(f"""
        import importlib.abc, importlib.machinery, os, sys, types
        sys.path.insert(0, {ROOT!r})

        ABSENT = ("wandb", "torchvision", "timm", "pytorch_metric_learning", "yourdfpy")

        class Stub(types.ModuleType):
            __path__ = []
            def __getattr__(self, name):
                if name.startswith("__"):
                    raise AttributeError(name)
                return type(name, (), {{}})

        class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
            def find_spec(self, fullname, path, target=None):
                if fullname.split(".")[0] in ABSENT:
                    return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            def create_module(self, spec):
                return Stub(spec.name)
            def exec_module(self, module):
                pass

        sys.meta_path.insert(0, Finder())
        import bevrender_amd.dropin
        bevrender_amd.dropin.install()

        os.chdir({REF!r})
        sys.path.insert(0, {REF!r})
        src = open(os.path.join({REF!r}, "train.py")).read().splitlines()
        block = "\\n".join(src[:26])                   # train.py:1-26, the import block
        assert "from model.bevrender import BEVRender" in block and "from loss.mse_loss import MSELoss" in block
        ns = {{}}
        exec(compile(block, "train.py[1:26]", "exec"), ns)
        mine = lambda o: o.__module__.startswith("bevrender_amd.")
        assert mine(ns["BEVRender"]), ns["BEVRender"].__module__
        for n in ("ContrastiveLoss", "LiftedStructureLoss", "TripletLossMetricLearning"):
            assert mine(ns[n]), (n, ns[n].__module__)
        for n in ("MSELoss", "L1Loss", "CrossEntropyLoss"):
            f = sys.modules[ns[n].__module__].__file__
            assert ns[n].__module__.startswith("loss.") and f.startswith({REF!r}), (n, f)
        import model.SCA_deform_attn as m
        assert m.__name__ == "bevrender_amd.model.SCA_deform_attn"
        print("SHIM-OK")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SHIM-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
