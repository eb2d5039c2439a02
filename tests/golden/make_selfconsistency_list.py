#!/usr/bin/env python3
"""The configuration list of the reference's self-consistency test (tests/test_general.py:116-390: `Test.setUp` fills `flow_inits` with
[[pdf_defs, flow_defs], kwargs] entries) as DATA: the reference's test module is imported, setUp() is run, and the resulting list is written
to tests/golden/selfconsistency_list.json.  Nothing of the module's text is stored.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_selfconsistency_list.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))


def jsonable(o):
    if isinstance(o, dict):
        return {str(k) if not isinstance(k, (int, str)) else k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    return o


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        spec = importlib.util.spec_from_file_location("ref_test_general", os.path.join(REF, "tests", "test_general.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        t = mod.Test()
        t.setUp()
    entries = []
    for (defs, kwargs) in t.flow_inits:
        # integer keys of options_overwrite (sub-pdf indices) survive JSON as strings: mark them
        kw = jsonable(kwargs)
        oo = kw.get("options_overwrite")
        if isinstance(oo, dict):
            kw["options_overwrite"] = {("#%d" % k if isinstance(k, int) else k): v for k, v in oo.items()}
        entries.append({"pdf_defs": defs[0], "flow_defs": defs[1], "kwargs": kw})
    out = {"source": "tests/test_general.py:116-390 (Test.setUp -> flow_inits) of thoglu/jammy_flows v1.1.0, dumped by make_selfconsistency_list.py",
           "samplesize": 10000, "conditional_input_dim_added": 2, "entries": entries}
    path = os.path.join(HERE, "selfconsistency_list.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("%d entries -> %s" % (len(entries), path))


if __name__ == "__main__":
    main()
