#!/bin/bash
# Damaged-input run of the texture readers (JPEG, EXR incl. PIZ / PXR24, PNG, TGA, PFM) under AddressSanitizer + UBSan, on the CPU:
# tools/fuzz/run.sh [SEED] [COUNT].  Files are made from small valid images by byte flips, truncation, spliced noise and header damage;
# every file must be read or rejected without a sanitizer report.
here="$(cd "$(dirname "$0")" && pwd)"; root="$here/../.."
seed="${1:-7}"; count="${2:-6000}"
work="$(mktemp -d)"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$root/pbrt-r3_amd/csrc/host" -I"$root/pbrt-r3_amd/csrc" -I"$root/include" \
    "$here/read_images_main.cpp" "$root"/pbrt-r3_amd/csrc/host/{pth_texture_image,pth_jpeg,pth_exr_codecs}.cpp -lz -o "$work/reader" || exit 1
python3 "$here/make_damaged_images.py" "$seed" "$work/files" "$count" || exit 1
cd "$work/files" && ls | grep "^f" | xargs -n 1000 "$work/reader"
status=$?
rm -rf "$work"
exit $status
