"""DESIGN.md's result tables are GENERATED from the tracked evidence under profiles/ (tools/design_tables.py): this test fails
when a number in them differs from the files, and when prose outside the generated block starts quoting tracked lines again."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_tables_are_what_profiles_say():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), "--check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-3000:]


def test_design_prose_quotes_no_tracked_line_numbers():
    """Numbers of the tracked bench lines live ONLY inside the generated block (round-2 review: DESIGN.md quoted a tracked
    line as 0.687 while the file said 0.600). Outside it the prose may cite experiments with their own profiles/ file."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import design_tables as dt
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert dt.BEGIN in text and dt.END in text
    a, b = text.index(dt.BEGIN), text.index(dt.END)
    prose = text[:a] + text[b:]
    assert not re.search(r"tracked (bench )?line[^\n]{0,40}\d", prose), "a tracked-line number is quoted outside the generated block"
    # every profiles/ file the document names exists (files of rounds 1-3 live under profiles/archive/)
    for name in set(re.findall(r"profiles/([A-Za-z0-9_.*/-]+)", text)):
        if "*" in name:
            import glob
            assert glob.glob(os.path.join(ROOT, "profiles", name)), "DESIGN.md names profiles/%s, which matches nothing" % name
        else:
            assert os.path.exists(os.path.join(ROOT, "profiles", name.rstrip(".,;:)"))), "DESIGN.md names profiles/%s, which does not exist" % name


def test_design_prose_fracs_are_the_tracked_lines():
    """Round-4 review: DESIGN.md's prose said "0.720 / 0.718" while the tracked C4 line said 0.698 / 0.706. Every roofline fraction the
    prose QUOTES (a number next to the word frac / fraction / roofline, outside the generated block) must be one the tracked lines of the
    current round carry — `roofline.frac` or `roofline.frame_frac` of some profiles/<tag>_bench_*.json, or of a stability run, to the
    printed precision — unless it is a stated TARGET (>=, <=, "target") or the sentence cites an experiment's own profiles/ file."""
    import glob
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import design_tables as dt
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = text.index(dt.BEGIN), text.index(dt.END)
    prose = text[:a] + text[b + len(dt.END):]
    tracked = set()
    for f in glob.glob(os.path.join(ROOT, "profiles", "%s_bench_*.json" % dt.TAG)):
        d = json.load(open(f))
        for k in ("frac", "frame_frac"):
            v = d.get("roofline", {}).get(k)
            if v is not None:
                tracked.update({"%.2f" % v, "%.3f" % v})
    sj = os.path.join(ROOT, "profiles", "%s_autotune_stability.json" % dt.TAG)
    assert tracked, "no tracked bench lines of round %s under profiles/" % dt.TAG
    bad = []
    for line in prose.splitlines():
        for m in re.finditer(r"(frac(?:tion)?|roofline)[^0-9\n]{0,24}(0\.\d{2,3})\b", line):
            lead = line[max(0, m.start(2) - 12):m.start(2)]
            if re.search(r"(≥|≤|>=|<=|target)", lead) or "profiles/" in line:
                continue
            if m.group(2) not in tracked:
                bad.append((m.group(2), line.strip()[:140]))
    assert not bad, "fractions quoted in DESIGN.md's prose that no tracked line of round %s carries: %r" % (dt.TAG, bad)
