"""The tuning-knob table of KNOBS.md, generated from the KNOBS registry in csrc/host_rt.hpp.
    python scripts/gen_knob_table.py          -> prints the markdown table
    python scripts/gen_knob_table.py --write  -> rewrites the block between <!-- knobs:begin --> and <!-- knobs:end --> in KNOBS.md"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rofl_project_code_amd", "csrc")
SRC = os.path.join(CSRC, "host_rt.hpp")                                               # the KNOBS registry
HOST_SOURCES = ["rofl_zk.hip", "host_rt.hpp", "host_msm.hpp", "host_prover.hpp", "host_verifier.hpp"]      # everything that may call knob()


def knobs():
    s = open(SRC).read()
    body = s[s.index("static const Knob KNOBS[] = {"):]
    body = body[:body.index("\n};")]
    return re.findall(r'\{"(ROFL_[A-Z0-9_]+)",\s*"([^"]*)",\s*"((?:[^"\\]|\\.)*)"\}', body)


def reads():
    """every name the sources pass to knob()"""
    s = "".join(open(os.path.join(CSRC, f)).read() for f in HOST_SOURCES)
    return sorted(set(re.findall(r'knob\("(ROFL_[A-Z0-9_]+)"\)', s)))


def table():
    rows = ["| variable | default | meaning |", "|---|---|---|"]
    for name, dflt, what in knobs():
        rows.append("| `%s` | %s | %s |" % (name, ("`%s`" % dflt) if dflt else "(unset)", what.replace("|", "\\|")))
    return "\n".join(rows)


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "KNOBS.md")
        d = open(p).read()
        a, b = d.index("<!-- knobs:begin -->") + len("<!-- knobs:begin -->"), d.index("<!-- knobs:end -->")
        open(p, "w").write(d[:a] + "\n" + t + "\n" + d[b:])
    else:
        print(t)
