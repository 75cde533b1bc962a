#!/usr/bin/env python3
"""Generate the log-format fixtures tests/golden/F9_logs_<system>.json by RUNNING the reference's own preset scripts
headless with ``--is_log_data`` / ``--is_print_sim_step`` on (build container only; /root/reference is read-only and
never copied).  A fixture holds data only: the argv used, the CSV text the reference wrote (20 header rows, the column
row, the data rows) and the console text it printed (the ``tabulate`` grids of ``print_sim_step``).

    python oracle/gen_log_fixtures.py

Recipe: SURVEY.md Appendix B stubs for the two GUI-only imports; each preset runs through ``runpy`` from a scratch
working directory so ``simdata/`` lands there.  Under NumPy >= 1.25 ``main_2tank.py`` dies in ``stage_obj`` on
``observation_target == []``; the generator wraps ``CtrlOptPred.__init__`` to hand the target over as the
``TargetArray`` ndarray subclass of gen_fixtures.py, which restores the old comparison without touching the arithmetic.
"""
import contextlib
import glob
import io
import json
import os
import runpy
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_fixtures import OUT, REF, TargetArray, import_reference  # noqa: E402

RUNS = {
    "3wrobot": ("main_3wrobot.py", ["--ctrl_mode", "MPC", "--t1", "0.06", "--Nactor", "3"]),
    "3wrobotNI": ("main_3wrobot_NI.py", ["--ctrl_mode", "MPC", "--t1", "0.06"]),
    "2tank": ("main_2tank.py", ["--ctrl_mode", "MPC", "--t1", "0.6", "--Nactor", "4"]),
}


def main():
    systems, simulator, controllers = import_reference()
    orig_init = controllers.CtrlOptPred.__init__

    def patched(self, *a, **k):
        if isinstance(k.get("observation_target"), np.ndarray):
            k["observation_target"] = TargetArray(k["observation_target"])
        return orig_init(self, *a, **k)

    controllers.CtrlOptPred.__init__ = patched
    os.makedirs(OUT, exist_ok=True)
    for name, (script, extra) in RUNS.items():
        argv = extra + ["--is_log_data", "1", "--is_visualization", "", "--is_print_sim_step", "1", "--Nruns", "2"]
        with tempfile.TemporaryDirectory() as tmp:
            cwd, old_argv = os.getcwd(), sys.argv
            os.chdir(tmp)
            sys.argv = [script] + argv
            buf = io.StringIO()
            died = None
            try:
                with contextlib.redirect_stdout(buf):
                    runpy.run_path(os.path.join(REF, "presets", script), run_name="__main__")
            except NameError as e:
                # the reference's headless reset between runs uses an undefined name (presets/main_3wrobot.py:461):
                # with --Nruns 2 both files get their header, run 1 is complete, then the script dies here
                died = repr(e)
            finally:
                os.chdir(cwd)
                sys.argv = old_argv
            files = sorted(glob.glob(os.path.join(tmp, "simdata", "*.csv")))
            csv_texts = [open(f, newline="").read() for f in files]
            names = [os.path.basename(f) for f in files]
        out = dict(system=name, script=script, argv=argv, csv_file_names=names, csv_texts=csv_texts,
                   stdout=buf.getvalue(), reference_died_with=died)
        path = os.path.join(OUT, f"F9_logs_{name}.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        print(f"wrote {path} ({os.path.getsize(path)} bytes; {len(files)} csv files, "
              f"{sum(t.count(chr(10)) for t in csv_texts)} csv lines)")


if __name__ == "__main__":
    main()
