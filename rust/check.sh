#!/usr/bin/env bash
# For a machine with cargo (this repository's build image has none):
#   rust/check.sh                                  cargo check of polymath-hip-sys and polymath-hip (nothing links: no GPU,
#                                                  no libpolymath_hip.so needed), with and without the bn254 feature
#   POLYMATH_REF=/path/to/sigma0-dev/polymath rust/check.sh
#                                                  ... then copies the reference, applies reference-patch/sigma0-polymath-hip.patch,
#                                                  points its `polymath-hip` dependency at this checkout and runs
#                                                  `cargo check --features hip` and (CPU path untouched) `cargo check`
#   POLYMATH_HIP_LIB_DIR=<repo>/polymath_amd POLYMATH_REF=... rust/check.sh --test
#                                                  ... and, on a box with an MI355X, the reference's own tests (tests/dummy.rs,
#                                                  tests/mimc.rs) through the GPU backend
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
cd "$here"
cargo check --workspace
cargo check --workspace --features polymath-hip/bn254
if [ -n "${POLYMATH_REF:-}" ]; then
    work="$(mktemp -d)"
    trap 'rm -rf "$work"' EXIT
    cp -r "$POLYMATH_REF" "$work/polymath"
    cd "$work/polymath"
    patch -p1 < "$here/reference-patch/sigma0-polymath-hip.patch"
    sed -i "s#path = \"../polymath-mi355x/rust/polymath-hip\"#path = \"$here/polymath-hip\"#" Cargo.toml
    cargo check                                   # the CPU crate is unchanged without the feature
    cargo check --features hip
    if [ "${1:-}" = "--test" ]; then
        : "${POLYMATH_HIP_LIB_DIR:?set POLYMATH_HIP_LIB_DIR to the directory that holds libpolymath_hip.so}"
        LD_LIBRARY_PATH="$POLYMATH_HIP_LIB_DIR:${LD_LIBRARY_PATH:-}" cargo test --release --features hip
    fi
fi
echo "rust/check.sh: ok"
