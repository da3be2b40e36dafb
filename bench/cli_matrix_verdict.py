"""One-line verdict of a main.py run for bench/cli_matrix.sh: losses from the stdout log, facts from the summary JSON."""
import json
import math
import re
import sys

text = open(sys.argv[2]).read()
losses = [float(m.group(1)) for m in re.finditer(r"step: \d+ loss: ([-+0-9.eE]+|nan|inf)", text)]
try:
    s = json.load(open(sys.argv[1]))
except Exception:  # noqa: BLE001
    s = {"summary": "missing"}
keep = {k: s.get(k) for k in ("images_per_sec_synchronised", "hip_graph", "hip_graph_failed", "library_fallbacks",
                              "reducer", "summary")}
keep = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in keep.items() if v not in (None, {}, False)}
finite_only = len(sys.argv) > 3 and sys.argv[3] == "finite-only"  # (too few steps to expect a lower loss)
ok = bool(losses) and all(math.isfinite(v) for v in losses) and (finite_only or losses[-1] < losses[0]) and "summary" not in keep
print(("OK " if ok else "CHECK ") + "loss %s -> %s " % (losses[0] if losses else None, losses[-1] if losses else None)
      + json.dumps(keep))
