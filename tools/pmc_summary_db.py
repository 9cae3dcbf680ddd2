#!/usr/bin/env python3
"""Mean counter value per dispatch, per kernel, from the rocpd SQLite file rocprofv3 --pmc writes (when it writes no CSV)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = next((t for t in tabs if t.lower() in ("counters_collection", "pmc_events")), None)
if view is None:
    print("tables:", tabs)
    sys.exit(0)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
print("# columns:", cols)
kn = next((c for c in cols if "kernel" in c.lower() and "name" in c.lower()), "name" if "name" in cols else None)
cn = next((c for c in cols if c.lower() == "counter_name"), None)
cv = next((c for c in cols if c.lower() in ("counter_value", "value")), None)
if not (kn and cn and cv):
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(list))
for k, c, v in db.execute(f"select {kn}, {cn}, {cv} from {view}"):
    acc[k][c].append(float(v))
for k, d in acc.items():
    print(str(k)[:100])
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
