"""Console / CSV log contract of the reference.

Mirrors ``rcognita/loggers.py:41-94`` (classes ``Logger3WRobot``, ``Logger3WRobotNI``, ``Logger2Tank`` with
``print_sim_step`` / ``log_data_row`` taking the same positional arguments) and the file layout the reference presets
write around them (``presets/main_3wrobot.py:328-362``): 20 ``key,value`` header rows, one column row, then one data
row per simulation step, through ``csv.writer`` - so a log written here is byte-compatible with one written by the
reference for the same numbers, and the batched runner logs env 0 in that format.  :func:`read_log` is the inverse
and returns the columns in the argument order of the reference animators' ``set_sim_data``
(``visuals.py:208, 458, 690``) for playback.  ``tests/test_loggers.py`` pins all of it on files and console text
captured from the reference's own preset scripts (``tests/golden/F9_logs_*.json``).
"""
from __future__ import annotations

import csv
from datetime import datetime

import numpy as np
from tabulate import tabulate

# header keys, in file order, after 'System' and 'Controller' (presets/main_3wrobot.py:343-360)
HEADER_KEYS = ("dt", "state_init", "is_est_model", "model_est_stage", "model_est_period_multiplier", "model_order",
               "prob_noise_pow", "Nactor", "pred_step_size_multiplier", "buffer_size", "stage_obj_struct", "R1_diag",
               "R2_diag", "Ncritic", "gamma", "critic_period_multiplier", "critic_struct", "actor_struct")
N_HEADER_ROWS = 2 + len(HEADER_KEYS)  # 20


class Logger:
    """Interface of the reference's loggers (loggers.py:17-35); concrete loggers are table-driven here."""
    columns: tuple = ()
    formats: tuple = ()
    n_state = 0        # leading state columns after t
    action_last = True  # action columns close the row (3wrobot, NI); 2tank logs the action array as one cell
    n_action = 0

    def _row(self, args):
        """positional args of print_sim_step/log_data_row -> flat list of cells in column order"""
        if not self.action_last:
            return list(args)
        *head, action = args
        a = np.ravel(np.asarray(action))
        return [*head, *[a[i] for i in range(self.n_action)]]

    def print_sim_step(self, *args):
        # a one-element action array (2tank's ``p``) is shown as its value: same text, no array->scalar coercion
        cells = [float(c[0]) if isinstance(c, np.ndarray) and c.size == 1 else c for c in self._row(args)]
        print(tabulate([list(self.columns), cells], floatfmt=self.formats, headers="firstrow", tablefmt="grid"))

    def log_data_row(self, datafile, *args):
        with open(datafile, "a", newline="") as outfile:
            csv.writer(outfile).writerow(self._row(args))


class Logger3WRobot(Logger):
    """print_sim_step(t, xCoord, yCoord, alpha, v, omega, stage_obj, accum_obj, action) (loggers.py:41-58)"""
    columns = ("t [s]", "x [m]", "y [m]", "alpha [rad]", "v [m/s]", "omega [rad/s]", "stage_obj", "accum_obj", "F [N]",
               "M [N m]")
    formats = ("8.3f", "8.3f", "8.3f", "8.3f", "8.3f", "8.3f", "8.1f", "8.1f", "8.3f", "8.3f")
    n_state, n_action = 5, 2
    playback_order = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)  # ts, xCoords, yCoords, alphas, vs, omegas, rs, accum_objs, Fs, Ms


class Logger3WRobotNI(Logger):
    """print_sim_step(t, xCoord, yCoord, alpha, stage_obj, accum_obj, action) (loggers.py:60-77)"""
    columns = ("t [s]", "x [m]", "y [m]", "alpha [rad]", "stage_obj", "accum_obj", "v [m/s]", "omega [rad/s]")
    formats = ("8.3f", "8.3f", "8.3f", "8.3f", "8.1f", "8.1f", "8.3f", "8.3f")
    n_state, n_action = 3, 2
    playback_order = (0, 1, 2, 3, 4, 5, 6, 7)  # ts, xCoords, yCoords, alphas, rs, accum_objs, vs, omegas


class Logger2Tank(Logger):
    """print_sim_step(t, h1, h2, p, stage_obj, accum_obj) (loggers.py:79-94); ``p`` is the action array itself, so the
    CSV cell reads ``[0.5]``"""
    columns = ("t [s]", "h1", "h2", "p", "stage_obj", "accum_obj")
    formats = ("8.1f", "8.4f", "8.4f", "8.4f", "8.4f", "8.2f")
    n_state, n_action = 2, 1
    action_last = False
    playback_order = (0, 1, 2, 3, 4, 5)  # ts, h1s, h2s, ps, rs, accum_objs


LOGGERS = {"3wrobot": Logger3WRobot, "3wrobotNI": Logger3WRobotNI, "2tank": Logger2Tank}


def datafile_names(data_folder, sys_name, ctrl_mode, Nruns, now=None):
    """One file per run: ``<folder>/<system>__<mode>__<YYYY-MM-DD>__<HHhMMmSSs>__runNN.csv``
    (presets/main_3wrobot.py:330-335)."""
    now = now or datetime.now()
    date, time = now.strftime("%Y-%m-%d"), now.strftime("%Hh%Mm%Ss")
    return [f"{data_folder}/{sys_name}__{ctrl_mode}__{date}__{time}__run{k + 1:02d}.csv" for k in range(Nruns)]


def write_header(datafile, sys_name, ctrl_mode, settings: dict, columns):
    """The 20 header rows + the column row (presets/main_3wrobot.py:340-362).  ``settings`` maps every key of
    :data:`HEADER_KEYS` to the object the preset holds; cells are ``str(obj)`` as in the reference."""
    with open(datafile, "w", newline="") as outfile:
        w = csv.writer(outfile)
        w.writerow(["System", sys_name])
        w.writerow(["Controller", ctrl_mode])
        for k in HEADER_KEYS:
            w.writerow([k, str(settings[k])])
        w.writerow(list(columns))


def _cell(v):
    v = v.strip()
    if v.startswith("[") and v.endswith("]"):  # 2tank action cell
        v = v[1:-1].split()[0]
    return float(v)


def read_log(datafile):
    """Parse a log written by :func:`write_header` + ``log_data_row`` (or by the reference).  Returns
    ``(header: dict, columns: list[str], data: float64 [rows, n_columns])``."""
    with open(datafile, newline="") as f:
        rows = list(csv.reader(f))
    header = {r[0]: (r[1] if len(r) > 1 else "") for r in rows[:N_HEADER_ROWS]}
    columns = rows[N_HEADER_ROWS]
    body = rows[N_HEADER_ROWS + 1:]
    data = np.array([[_cell(c) for c in r] for r in body], dtype=np.float64).reshape(len(body), len(columns))
    return header, columns, data


def playback_args(datafile):
    """Columns of a log as the positional arguments of the matching reference animator's ``set_sim_data``
    (``Animator3WRobot`` visuals.py:208, ``Animator3WRobotNI`` :458, ``Animator2Tank`` :690)."""
    header, _, data = read_log(datafile)
    lg = LOGGERS[header["System"]]
    return tuple(data[:, c] for c in lg.playback_order)
