"""Run a script (or `-m module`) of this repo against a VARIANT build of the native library:

    python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_<tag>.so bench.py --no-cpu-baseline --no-e2e
    python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_<tag>.so -m pytest tests/test_gpu_parity.py -m gpu -q

A/B tooling for measured experiments.  The product loader (codenet_amd/_native.py) has no
environment override: a variant -- including the deliberately wrong `make diag` builds -- can only be selected by
this explicit runner, which calls _native.use_library(path) before anything loads the library."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    so = sys.argv[1]
    if not os.path.isabs(so):
        so = os.path.join(ROOT, so)
    from codenet_amd import _native
    _native.use_library(so)
    print("with_lib: %s" % _native.SO_PATH, file=sys.stderr)
    if sys.argv[2] == "-m":
        mod = sys.argv[3]
        sys.argv = [mod] + sys.argv[4:]
        runpy.run_module(mod, run_name="__main__", alter_sys=True)
    else:
        script = sys.argv[2]
        sys.argv = [script] + sys.argv[3:]
        sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
        runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
