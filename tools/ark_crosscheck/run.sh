#!/usr/bin/env bash
# One command for a machine that has cargo (this repo's build image does not):
#   POLYMATH_REF=/path/to/sigma0-dev/polymath ./run.sh [--bench]
# 1. emit   : the reference's own setup/prove on tests/dummy.rs and tests/mimc.rs shapes, test_rng seeds
#             -> tests/golden/ref_dummy.json, ref_inputs11.json, ref_inputs0.json, ref_mimc322.json   (consumed by tests/test_reference_fixtures.py)
# 2. verify : this repo's 24 BLS12-381 golden proofs (m0 = 1, 2, 3 and 12) through the reference's Polymath::verify -> must print "24 / 24 accepted"
# 3. --bench: ark-ec msm_unchecked / ark-poly fft at 2^20 .. 2^24 -> tests/golden/ref_cpu_baseline.json
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
ref="${POLYMATH_REF:?set POLYMATH_REF to a checkout of sigma0-dev/polymath}"
sed -i "s#^sigma0-polymath = { path = \"[^\"]*\"#sigma0-polymath = { path = \"$ref\"#" "$here/Cargo.toml"
cd "$here"
cargo run --release -- emit "$root/tests/golden"
cargo run --release -- verify "$root/tests/golden/proofs.json"
if [ "${1:-}" = "--bench" ]; then
    RAYON_NUM_THREADS="${RAYON_NUM_THREADS:-$(nproc)}" cargo run --release -- bench "$root/tests/golden/ref_cpu_baseline.json"
fi
echo "now run:  python -m pytest tests/test_reference_fixtures.py -q        (CPU: pins the oracle;  -m gpu: pins the HIP path)"
