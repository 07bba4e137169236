"""Register / spill summary of every kernel in build/asm (make -C open_l2o_amd/csrc asm: the MAIN translation unit; the kernels of
csrc/l2o_ilp_kernels.h: scripts/tu_regs.sh ... -- -mllvm -amdgpu-sched-strategy=max-ilp, or scripts/so_regs.py on the built library)."""
import re, sys
args = [a for a in sys.argv[1:] if not a.startswith('--')]
s = open(args[0] if args else 'build/asm/l2o_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
only_spills = '--spills' in sys.argv
for b in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    ag = int(b.split('\n')[0])
    vg = int(re.search(r'\.vgpr_count:\s+(\d+)', b).group(1))
    sp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', b).group(1))
    sc = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', b).group(1))
    if (sp or sc) if only_spills else True:
        print(f"{name[:80]:80s} agpr {ag:3d} total {vg:3d} spill {sp:3d} scratch {sc}")
