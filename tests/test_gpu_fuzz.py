"""A slice of tools/fuzz_warp.py in the suite: random frame sizes, meshes and motion strengths through cell table + plan + warp kernel
against the C oracle, bit for bit (the tool ran 16,000 small and 1,500 large -- up to 1280 x 720 -- cases on the final kernels of round 2 and 13,000 + 2,600 on those of round 3, 0 mismatches; this keeps 120 as a regression net)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def test_random_geometries_bit_exact():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuzz_warp
    assert fuzz_warp.run(120, seed=20261002, verbose=False) == 0
