"""Runs on the GPU box (profiles/run_profiles.sh): a rocprofv3 --pmc pass leaves a ~45 MB rocpd database, gpurun merges at most 64 MiB back.
Replaces every *.db under the given directories by agg.json = {kernel name: {counter: [dispatches, mean value per dispatch]}} -- all that
profiles/make_profiles.py reads from a counter pass."""
import glob
import json
import os
import sqlite3
import sys

for d in sys.argv[1:]:
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        out = {}
        for name, ctr, n, mean in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                              "group by kernel_name, counter_name"):
            out.setdefault(name, {})[ctr] = [n, mean]
        con.close()
        with open(os.path.join(d, "agg.json"), "w") as f:
            json.dump(out, f)
        os.remove(db)
