#!/usr/bin/env python3
"""Make a collection (tools/collect_r06.sh <tag> on the GPU box -> gpurun_out/<tag>/) the set profiles/ is judged by: copies
its summaries to profiles/<tag>_*, writes profiles/LATEST.json (tag, the git head the collection ran on, the files by kind) -
the manifest bench.py's roofline object and tools/check_design_numbers.py select files through, instead of a glob over hundreds
of files - and checks DESIGN.md's quoted trace numbers against the adopted trace summary.
usage: tools/adopt_profiles.py <tag>"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    man = json.load(open(os.path.join(src, 'LATEST.json')))
    for kind, rel in man['files'].items():
        name = os.path.basename(rel)[len(tag) + 1:]
        shutil.copy(os.path.join(src, name), os.path.join(ROOT, rel))
    for f in sorted(os.listdir(src)):                       # the bench lines and logs of the same collection
        if f.endswith(('_bench_line.json', '.log')) or f == 'bench_line.json':
            shutil.copy(os.path.join(src, f), os.path.join(ROOT, 'profiles', '%s_%s' % (tag, f)))
    if man.get('head') in (None, 'unknown'):
        man['head'] = subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip() + ' (adopted at)'
    json.dump(man, open(os.path.join(ROOT, 'profiles', 'LATEST.json'), 'w'), indent=1)
    print('adopted', tag, 'head', man['head'], sorted(man['files']))
    sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_design_numbers.py')]).returncode)


if __name__ == '__main__':
    main()
