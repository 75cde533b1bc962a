"""Log contract: rcognita_amd.loggers + the preset header against files and console text the
reference's own preset scripts produced (tests/golden/F9_logs_*.json, made by oracle/gen_log_fixtures.py).  CPU only."""
import contextlib
import csv
import io
import json
import os
import re
from datetime import datetime

import numpy as np
import pytest

from rcognita_amd import loggers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SYSTEMS = ["3wrobot", "3wrobotNI", "2tank"]


def _fixture(name):
    with open(os.path.join(GOLDEN, f"F9_logs_{name}.json")) as f:
        return json.load(f)


def _ref_rows(text):
    rows = list(csv.reader(io.StringIO(text, newline="")))
    return rows[:loggers.N_HEADER_ROWS], rows[loggers.N_HEADER_ROWS], rows[loggers.N_HEADER_ROWS + 1:]


def _args_of_row(name, row):
    """CSV cells of one reference data row -> the positional arguments the reference loop passes to its logger"""
    if name == "2tank":
        t, h1, h2, p, so, ao = row
        return (float(t), float(h1), float(h2), np.array([float(p.strip("[]"))]), float(so), float(ao))
    vals = [float(c) for c in row]
    return (*vals[:-2], np.array(vals[-2:]))


@pytest.mark.parametrize("name", SYSTEMS)
def test_header_rows_match_reference_bytes(name, tmp_path):
    """20 header rows + the column row written from the SAME argv, byte for byte."""
    from rcognita_amd.presets import build_parser

    fx = _fixture(name)
    args = build_parser(name).parse_args(fx["argv"])
    settings = {k: getattr(args, k) for k in loggers.HEADER_KEYS if k != "state_init"}
    settings["state_init"] = np.array([eval(v.replace("pi", str(np.pi))) for v in args.state_init])
    f = tmp_path / "h.csv"
    loggers.write_header(str(f), name, args.ctrl_mode, settings, loggers.LOGGERS[name].columns)
    ours = f.read_bytes().decode()
    ref_lines = fx["csv_texts"][0].split("\r\n")
    n = loggers.N_HEADER_ROWS + 1
    assert ours.split("\r\n")[:n] == ref_lines[:n]
    assert ours == fx["csv_texts"][1]  # the reference's run-02 file is header only (it dies before run 2)
    assert loggers.N_HEADER_ROWS == 20


@pytest.mark.parametrize("name", SYSTEMS)
def test_data_rows_and_console_match_reference_bytes(name, tmp_path):
    """Feed the reference's logged numbers back through this build's logger: identical CSV bytes and identical
    ``tabulate`` console grids."""
    fx = _fixture(name)
    _, cols, body = _ref_rows(fx["csv_texts"][0])
    lg = loggers.LOGGERS[name]()
    assert list(lg.columns) == cols
    f = tmp_path / "d.csv"
    f.write_text("")
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        for row in body:
            a = _args_of_row(name, row)
            lg.print_sim_step(*a)
            lg.log_data_row(str(f), *a)
    ref_body = "\r\n".join(fx["csv_texts"][0].split("\r\n")[loggers.N_HEADER_ROWS + 1:])
    assert f.read_bytes().decode() == ref_body
    # console: everything the reference printed between the 'Logging data to' lines and the run-done banner
    ref_out = fx["stdout"]
    start = ref_out.index("+--")
    end = ref_out.index("....")
    assert out.getvalue() == ref_out[start:end]
    assert ".....................................Run  1 done....................................." in ref_out


def test_datafile_names_pattern():
    names = loggers.datafile_names("simdata", "3wrobot", "MPC", 3, now=datetime(2026, 10, 3, 11, 10, 25))
    assert names == [f"simdata/3wrobot__MPC__2026-10-03__11h10m25s__run{k:02d}.csv" for k in (1, 2, 3)]
    for name in SYSTEMS:
        for fn in _fixture(name)["csv_file_names"]:
            assert re.fullmatch(rf"{name}__MPC__\d{{4}}-\d\d-\d\d__\d\dh\d\dm\d\ds__run\d\d\.csv", fn)


@pytest.mark.parametrize("name", SYSTEMS)
def test_read_log_and_playback_order(name, tmp_path):
    fx = _fixture(name)
    f = tmp_path / fx["csv_file_names"][0]
    f.write_bytes(fx["csv_texts"][0].encode())
    header, cols, data = loggers.read_log(str(f))
    assert header["System"] == name and header["Controller"] == "MPC" and len(header) == 20
    _, _, body = _ref_rows(fx["csv_texts"][0])
    assert data.shape == (len(body), len(cols))
    assert np.all(np.diff(data[:, 0]) > 0)  # time column
    pb = loggers.playback_args(str(f))
    # set_sim_data argument order (visuals.py:208, 458, 690) is the CSV column order for all three systems
    assert len(pb) == len(cols)
    for k, col in enumerate(pb):
        np.testing.assert_array_equal(col, data[:, k])
    if name == "2tank":
        assert np.all((pb[3] >= 0) & (pb[3] <= 1))  # the '[p]' cells parsed as numbers
