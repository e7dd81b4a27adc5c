"""Background HBM traffic for tests/test_gpu_repro.py: one process that keeps GPU 0's memory system busy -- write-only fills, read + write
axpy passes and element-wise adds over 3 x 256 MB, back to back through the C ABI -- until the stop file appears (or the deadline passes).
r04's k-split hazard (an inline-asm store whose data registers the compiler reused) never showed with the GPU to itself and
showed in 5 - 8 of 30 captured runs once another process's traffic backed the store path up: the hand-off tests run beside two of these.
usage: hbm_stream_worker.py READY_FILE STOP_FILE [DEADLINE_S]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ready, stop = Path(sys.argv[1]), Path(sys.argv[2])
    deadline = time.time() + (float(sys.argv[3]) if len(sys.argv) > 3 else 120.0)
    from taper_amd import hip
    ctx = hip.Ctx(0)
    n = 64 << 20                                    # 256 MB per buffer: past the 256 MB Infinity Cache in every pass
    a, b, c = ctx.empty(n), ctx.empty(n), ctx.empty(n)
    ctx.call("th_fill_f32", a, 1.0, n)
    ctx.call("th_fill_f32", b, 2.0, n)
    ctx.sync()
    ready.write_text("ready")
    passes = 0
    while not stop.exists() and time.time() < deadline:
        for _ in range(8):                          # ~1 ms of queued work between looks at the stop file
            ctx.call("th_fill_f32", c, float(passes & 7), n)
            ctx.call("th_axpy", 0.5, a, b, n)
            ctx.call("th_add", a, b, c, n)
            passes += 1
        ctx.sync()
    ctx.close()


if __name__ == "__main__":
    main()
