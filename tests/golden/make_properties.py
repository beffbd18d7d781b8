#!/usr/bin/env python3
"""Extract the interface definition of the reference components (class attributes and the three
property dictionaries of RRTMGLongwave / RRTMGShortwave / Instellation, constructor keyword defaults) into
tests/golden/reference_interface.json by parsing the reference source with `ast` (sympl is not installed,
so the modules cannot be imported).  Run in the build container only."""
import ast
import json
import os

REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_interface.json")

res = {}
for key, path, cls in (("RRTMGShortwave", "climt/_components/rrtmg/sw/component.py", "RRTMGShortwave"),
                       ("RRTMGLongwave", "climt/_components/rrtmg/lw/component.py", "RRTMGLongwave"),
                       ("Instellation", "climt/_components/instellation/component.py", "Instellation"),
                       ("BergerSolarInsolation", "climt/_components/berger_solar_insolation.py", "BergerSolarInsolation"),
                       ("SlabSurface", "climt/_components/slab_surface.py", "SlabSurface")):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    entry = {}
    for st in node.body:
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
            try:
                entry[st.targets[0].id] = ast.literal_eval(st.value)
            except ValueError:
                pass
        if isinstance(st, ast.FunctionDef) and st.name == "__init__":
            names = [a.arg for a in st.args.args][1:]
            defaults = [ast.literal_eval(d) for d in st.args.defaults]
            entry["__init__"] = dict(zip(names[len(names) - len(defaults):], defaults))
    res[key] = entry
tree = ast.parse(open(os.path.join(REF, "climt/_components/rrtmg/rrtmg_common.py")).read())
res["options"] = {st.targets[0].id: ast.literal_eval(st.value) for st in tree.body if isinstance(st, ast.Assign)}
json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
print("wrote", OUT, {k: list(v) for k, v in res.items() if k != "options"})
