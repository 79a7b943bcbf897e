"""One line per bench JSON file: value, p50, iterations per model step, cell us (tools/r05 batches)."""
import json
import sys

for p in sys.argv[1:]:
    try:
        with open(p) as f:
            txt = f.read().strip().splitlines()
        d = json.loads(txt[-1])
    except Exception as e:
        print(f"{p}: unreadable ({e})")
        continue
    r = d.get("roofline", {})
    st = d.get("stage_ms_per_model_step", {})
    print(f"{p.split('/')[-1]:<44} value {d.get('value'):>9} p50 {d.get('latency_ms', {}).get('p50_model_chunk')} "
          f"iters/step {st.get('decode_iters')} cell_us {r.get('launch_us')} ev {r.get('launch_us_events')} "
          f"tok_eq {d.get('tokens_equal')} prof_value {r.get('value_profiled')}")
