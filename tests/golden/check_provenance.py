#!/usr/bin/env python3
"""Provenance of tests/golden/survey_known_answers.json.

The reference's action / sampler translation units need Eigen and GSL, which this image lacks, so no script in this
repository can regenerate the vectors by running the reference.  They were recorded (printf %.17g) from the compiled
reference during the survey and are written down in SURVEY.md section 8(c); the JSON is a transcription of those
numbers.  This script checks the transcription: every floating-point value of the JSON must occur in SURVEY.md to at
least 12 significant digits (SURVEY.md prints some of them with %.17g, some rounded), integers verbatim.

    python tests/golden/check_provenance.py        # exit code 0 = every value found
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def numbers(node, path=""):
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("params", "_provenance"):
                continue
            yield from numbers(v, f"{path}/{k}")
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from numbers(v, f"{path}[{i}]")
    elif isinstance(node, bool):
        return
    elif isinstance(node, (int, float)):
        yield path, node


def main():
    survey = open(os.path.join(ROOT, "SURVEY.md")).read()
    found = [float(t) for t in re.findall(r"(?<![\w.])[-+]?\d+\.\d+(?:[eE][-+]?\d+)?|(?<![\w.])[-+]?\d+(?![\w.])", survey)]
    golden = json.load(open(os.path.join(HERE, "survey_known_answers.json")))
    missing = []
    for path, v in numbers(golden):
        ok = any(abs(v - f) <= 1e-12 * max(1.0, abs(v)) for f in found) if isinstance(v, float) else float(v) in found
        if not ok:
            missing.append((path, v))
    for path, v in missing:
        print(f"not in SURVEY.md: {path} = {v!r}")
    print(f"{sum(1 for _ in numbers(golden)) - len(missing)} values traced to SURVEY.md, {len(missing)} not found")
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
