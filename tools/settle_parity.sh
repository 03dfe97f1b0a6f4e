#!/bin/bash
# The ONE command to run the day a box has a real OpenCV (`python -c "import cv2"` works): settles SURVEY.md 8(c)'s "parity unpinned"
# for the six OpenCV calls the path restates (findHomography, warpPerspective, perspectiveTransform, remap -- mfs.py:1041-1069;
# resize -- mfs.py:1150; medianBlur -- mfs.py:359).
#
#     bash tools/settle_parity.sh [report.json]           (MESHFLOW_REFERENCE_DIR=/path/to/reference adds the reference's own loop)
#
# Runs tests/test_cv2_crosscheck.py (CPU only, ~1 min) and prints a JSON verdict:
#   "settled": true   every check passed in the mode the installed version calls for -- bit-exact for OpenCV 4.5 ... 4.10 (the
#                     fixed-point kernels the oracle models), <= 1 LSB / 1e-4 (BASELINE.json's bars) for any other version, with the
#                     count of values that are not bit-equal reported per check;
#   "settled": false  a check failed (pytest's output above says which), or there is no cv2 on this box ("reason").
# With the GPU present, `python -m pytest tests -m gpu -q` then ties the HIP kernels to the same oracle bit for bit.
cd "$(dirname "$0")/.."
REPORT=${1:-parity_report.json}
if ! python -c "import cv2" 2>/dev/null; then
  echo '{"settled": false, "reason": "no cv2 on this box (python -c \"import cv2\" fails): nothing was compared"}' | tee "$REPORT"
  exit 3
fi
rm -f "$REPORT.checks"
MESHFLOW_PARITY_REPORT="$REPORT.checks" python -m pytest tests/test_cv2_crosscheck.py -q -rs -p no:cacheprovider
RC=$?
python - "$REPORT" "$RC" <<'PY'
import json, sys
path, rc = sys.argv[1], int(sys.argv[2])
try:
    rep = json.load(open(path + '.checks'))
except (OSError, ValueError):
    rep = {'checks': []}
worst_px = max((c['max_abs'] for c in rep['checks'] if c['kind'] == 'pixels'), default=None)
worst_xy = max((c['max_abs'] for c in rep['checks'] if c['kind'] == 'coordinates'), default=None)
verdict = {'settled': rc == 0 and len(rep['checks']) > 0, 'pytest_exit_code': rc, 'cv2_version': rep.get('cv2_version'),
           'modelled_range': rep.get('modelled_range'), 'assertion_mode': rep.get('assertion_mode'),
           'checks_run': len(rep['checks']), 'checks_not_bit_equal': sum(1 for c in rep['checks'] if c['mismatching']),
           'worst_pixel_difference_lsb': worst_px, 'worst_coordinate_difference': worst_xy, 'checks': rep['checks']}
json.dump(verdict, open(path, 'w'), indent=1)
print(json.dumps({k: v for k, v in verdict.items() if k != 'checks'}))
PY
exit $RC
