"""Many proofs on the same contexts: nothing may grow per proof (tools/soak.py, short form).  The lazy stage timers of pm_host_prove
once leaked a helper context's events into its owner's pool -- 40 KB of host heap per sharded proof, invisible to every parity test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hundreds_of_proofs_leave_hbm_and_rss_flat():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--log-constraints", "14", "--proofs", "200", "--ranks", "4",
                        "--sharded-proofs", "200", "--tolerance-mb", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().split("\n")[-1])
    assert rec["ok"] and all(leg.get("hbm_growth_mb", leg.get("second_half_hbm_growth_mb", 0.0)) <= 3.0 for leg in rec["legs"]), rec
