#!/usr/bin/env python
"""Per-kernel PMC averages from rocprofv3 (rocpd sqlite) counter-collection runs."""
import re, sqlite3, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void lasr::", "").replace("lasr::", "")
    return n[:80]

def main(paths, filt=None):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        for name, did, cn, cv in cur.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
            k = short(name)
            if filt and filt not in k: continue
            a = agg[k][cn]; a[0] += cv; a[1].add(did)
    for k, cs in agg.items():
        print(k)
        for cn, (tot, d) in sorted(cs.items()):
            print(f"   {cn:<28} per-dispatch {tot/len(d):>16.1f}   (dispatches {len(d)})")

if __name__ == "__main__":
    args = sys.argv[1:]
    filt = None
    if "--filter" in args:
        i = args.index("--filter"); filt = args[i+1]; del args[i:i+2]
    main(args, filt)
