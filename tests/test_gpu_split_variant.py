"""The experimental eight-wave MFCC-39 kernel (mfcc_amd/csrc/split_kernel.h, built as a library of its own by
`make split`) gives the product kernel's rows bit for bit: tools/split_check.py, run as a child of the fork
server because it loads two builds of the library in one process."""
import multiprocessing as mp
import os
import sys

import pytest

from rank_worker import run_command

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_split_kernel_rows_equal_the_product_kernels():
    if not os.path.exists(os.path.join(ROOT, "build", "variants", "libmfcc_split.so")):
        pytest.skip("build/variants/libmfcc_split.so not built (make split)")
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    p = ctx.Process(target=run_command, args=([sys.executable, os.path.join(ROOT, "tools", "split_check.py")], ROOT, q))
    p.start()
    rc, out, err = q.get(timeout=580)
    p.join(60)
    assert rc == 0 and out.strip().endswith("OK"), f"rc {rc}\n{out[-3000:]}\n{err[-3000:]}"
    assert out.count("bit-identical") == 7
