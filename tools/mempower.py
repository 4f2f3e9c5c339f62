"""Socket power beside tools/mempower.hip's streaming kernels (a child process; the parent only samples the board)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import BoardSampler
exe = "/tmp/mempower"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "mempower.hip"), "-o", exe])
names = {0: "read 4 : write 1 through registers", 1: "read 4 : write 1 staged by LDS-DMA", 2: "read only", 3: "copy 1 : 1"}
for mode in (0, 1, 2, 3):
    p = subprocess.Popen([exe, str(mode), "4"], stdout=subprocess.PIPE, text=True)
    time.sleep(1.5)                                   # allocation, fill, clock transient
    s = BoardSampler(period_s=0.02); s.start(); time.sleep(2.0); tel = s.stop()
    out = p.communicate()[0].strip()
    print("%-38s %s | power %s W (max %s, cap %s) | SMU clock %s MHz" % (names[mode], out, tel.get("power_w"), tel.get("power_w_max"),
                                                                   tel.get("power_cap_w"), tel.get("gfx_mhz_smi")))
    time.sleep(1.0)
