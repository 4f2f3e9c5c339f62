"""An instruction price list at the power cap: tools/valu_power_probe.hip's mixes one at a time, each for a few seconds, with the
board's socket power and SMU clock sampled beside it (bench.py's BoardSampler).  Dynamic energy per launch = (power - idle power)
x time per launch; the price of an extra instruction = the difference to the plain-FMA mix / how many of them a launch executes
(mix descriptions say how many per 512 packed FMAs; a launch = 1024 workgroups x 4 waves x 256 tiles x 512 FMAs).

    hipcc --offload-arch=gfx950 -O3 -w tools/valu_power_probe.hip -o /tmp/valu_power_probe
    python3 tools/price_list.py [/tmp/valu_power_probe] [seconds=2.5]

The parent process never touches HIP: it samples amdsmi and starts the probe as a child per mix."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SXFIR_NO_TORCH"] = "1"
from bench import BoardSampler

exe = sys.argv[1] if len(sys.argv) > 1 else "/tmp/valu_power_probe"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
idle = BoardSampler(period_s=0.05); idle.start(); time.sleep(1.5); idle_w = idle.stop().get("power_w")
print("# idle socket power %s W" % idle_w)
MIXES = [1, 0, 6, 16, 4, 8, 17, 18, 19, 20, 21, 15, 14]
EXTRA = {6: ("ds_read_b128", 39), 16: ("ds_read_b128", 24), 4: ("ds_read_b128", 47), 17: ("v_cvt_f32_i32", 128), 18: ("v_mov_b32", 128),
         19: ("v_pk_add_f32", 128), 20: ("ds_write_b128", 16), 21: ("v_cndmask_b32", 128), 15: ("v_mov_b32_dpp wave_shl:1", 128)}
base = None
WAVE_TILES = 1024 * 4 * 256            # wave-tiles per launch, 512 packed FMAs each
for m in MIXES:
    smp = BoardSampler(period_s=0.02)
    p = subprocess.Popen([exe, str(m), str(secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    time.sleep(1.0)                    # the child's start-up and settle launches
    smp.start(0.0)
    out = p.communicate()[0]
    tel = smp.stop()
    mm = re.search(r"^(.*?)\s+([0-9.]+) ms per launch.*in-kernel clock (\d+) MHz", out, re.M)
    if not mm:
        print("mix %d: no result: %s" % (m, out[-200:])); continue
    name, ms, mhz = mm.group(1).strip(), float(mm.group(2)), int(mm.group(3))
    w = tel.get("power_w")
    line = "mix %2d %-80s %.4f ms | %s W | in-kernel %d MHz, SMU %s MHz" % (m, name[:80], ms, w, mhz, tel.get("gfx_mhz_smi"))
    if w and idle_w:
        e = (w - idle_w) * ms * 1e-3
        line += " | dynamic energy %.4f J per launch" % e
        if m == 1:
            base = e
            line += " = %.1f pJ per packed FMA lane-pair x 64 lanes (wave instruction: %.2f nJ)" % (e / (WAVE_TILES * 512 * 64) * 1e12, e / (WAVE_TILES * 512) * 1e9)
        elif base and m in EXTRA:
            what, per = EXTRA[m]
            line += " | one %s (wave instruction) = %.2f nJ = %.2f packed FMAs" % (what, (e - base) / (WAVE_TILES * per) * 1e9, (e - base) / per / (base / 512))
        elif base and m == 0:
            line += " | VGPR instead of SGPR tap: x %.3f per FMA" % (e / base)
    print(line, flush=True)
