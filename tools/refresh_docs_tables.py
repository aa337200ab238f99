#!/usr/bin/env python3
"""Replaces README.md's results table with tools/readme_table.py's output (numbers from profiles/r06_bench*.json)."""
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import readme_table  # noqa: E402

buf = io.StringIO()
with redirect_stdout(buf):
    readme_table.main()
table = buf.getvalue().split("\n\n")[0].rstrip("\n")
p = os.path.join(ROOT, "README.md")
s = open(p).read()
i = s.index("| workload (24 heads unless noted, D = 128) | K5 operands |")
j = s.index("\n\n", i)
open(p, "w").write(s[:i] + table + s[j:])
print(buf.getvalue().split("\n\n")[1])
