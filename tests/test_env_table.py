"""INTEGRATION.md's table of environment switches against the sources: every variable the product reads (getenv in
phnrec_amd/csrc, os.environ in phnrec_amd/*.py) has a row, and every row names a variable something reads."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_by_sources():
    names = set()
    for pat in ("phnrec_amd/csrc/*.cpp", "phnrec_amd/csrc/*.hip", "phnrec_amd/csrc/*.h", "phnrec_amd/csrc/host/*.cpp",
                "phnrec_amd/csrc/host/*.h", "include/*.h"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            names |= set(re.findall(r'getenv\("([A-Za-z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(ROOT, "phnrec_amd", "*.py")):
        src = open(f).read()
        names |= set(re.findall(r'environ(?:\.get|\.setdefault|\.pop)?[(\[]\s*"([A-Za-z0-9_]+)"', src))
        names |= set(re.findall(r'"([A-Z][A-Z0-9_]+)" (?:not )?in os\.environ', src))
    return names


def _rows_of_the_table():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## Environment switches"):]
    names = set()
    for line in sec.splitlines():
        if line.startswith("| `"):
            first = line.split("|")[1]
            names |= {n.split("=")[0] for n in re.findall(r"`([A-Za-z0-9_=,./…]+)`", first)}
    return names


def test_every_environment_variable_read_is_documented_and_every_row_is_read():
    read, rows = _read_by_sources(), _rows_of_the_table()
    assert read - rows == set(), "read by the sources, missing from INTEGRATION.md's table: %s" % sorted(read - rows)
    assert rows - read == set(), "rows of INTEGRATION.md's table that nothing reads: %s" % sorted(rows - read)
    assert len(read) >= 15
