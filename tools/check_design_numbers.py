#!/usr/bin/env python3
"""DESIGN.md may quote the step's trace only as the adopted collection says it (section 3 quoted a
superseded trace - 96.9 % busy / 1.38 in flight / 3.1 % idle - beside a final tree at 91.5 / 1.26 / 8.5).  DESIGN.md carries ONE
machine-checked line,
    <!-- trace: busy NN.N % | in flight N.NN | idle N.N % | source profiles/<file> -->
and this script (run by tools/adopt_profiles.py and by tests/test_experiments_cpu.py) fails unless source is the trace summary
profiles/LATEST.json names and the three numbers are the ones in that file."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def trace_numbers(path):
    txt = open(path).read()
    busy = re.search(r'busy\(union\)\s+[\d.]+ ms \(([\d.]+)%\)', txt)
    conc = re.search(r'avg concurrency ([\d.]+)', txt)
    idle = re.search(r'idle gaps: \d+, [\d.]+ ms = ([\d.]+)% of the window', txt)
    return float(busy.group(1)), float(conc.group(1)), float(idle.group(1))


def check():
    man = json.load(open(os.path.join(ROOT, 'profiles', 'LATEST.json')))
    rel = man['files']['trace_summary_4lanes']
    want = trace_numbers(os.path.join(ROOT, rel))
    m = re.search(r'<!-- trace: busy ([\d.]+) % \| in flight ([\d.]+) \| idle ([\d.]+) % \| source (\S+) -->', open(os.path.join(ROOT, 'DESIGN.md')).read())
    if not m:
        return 'DESIGN.md has no "<!-- trace: busy .. | in flight .. | idle .. | source .. -->" line'
    got = (float(m.group(1)), float(m.group(2)), float(m.group(3)))
    if m.group(4) != rel:
        return 'DESIGN.md quotes %s, the adopted trace summary is %s' % (m.group(4), rel)
    if got != want:
        return 'DESIGN.md quotes busy / in flight / idle = %s, %s says %s' % (got, rel, want)
    return None


if __name__ == '__main__':
    err = check()
    print('DESIGN.md trace numbers:', err or 'match profiles/LATEST.json')
    sys.exit(1 if err else 0)
