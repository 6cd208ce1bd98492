#!/usr/bin/env python3
"""Run an UNCHANGED reference-style caller (the reference's own test_last.py, or a script written against its module
names) on the MI355X path:

    python aa-clip-iqm_amd/run_reference_script.py /path/to/reference/test_last.py --dataset MVTec ...

Why a launcher: `python /path/to/reference/test_last.py` puts the script's directory at sys.path[0], in front of
PYTHONPATH.  The reference's `model/` has no __init__.py (a namespace portion: the build's regular package wins), but
its `dataset/` is a regular package and `forward_utils.py` / `utils.py` are plain modules, so those three would resolve
to the reference and the caller would mix the build's model with the reference's torch `calculate_similarity_map`
(which imports kornia).  Here the build's directory goes FIRST and the script's directory second, then the script
runs as __main__ (runpy, nothing is exec'd): model, dataset, forward_utils and utils all resolve to the build.
tests/test_host_cpu.py::test_documented_import_resolution pins both orders.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print(f"run_reference_script: {script} is not a file", file=sys.stderr)
        return 2
    here = os.path.dirname(os.path.abspath(__file__))
    sdir = os.path.dirname(script)
    sys.path[:] = [here] + [p for p in sys.path if os.path.abspath(p or ".") not in (here, sdir)] + [sdir]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
