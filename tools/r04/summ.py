"""One line per bench JSON: value, p50, host microseconds per model step, overlap probe, self-check, cell launch time."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
        pr = d["per_rank"][0]
        print(f.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "host", pr["host_us_per_model_step"],
              "overlap", pr.get("overlap_probe"), "eq", d.get("tokens_equal"), "cell_us", d["roofline"]["launch_us"])
    except Exception as e:
        print(f, "ERR", e)
