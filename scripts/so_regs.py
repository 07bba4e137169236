"""Register / spill / LDS summary of the kernels INSIDE a built libl2o_hip.so (no -save-temps rebuild needed):
   python scripts/so_regs.py [lib.so] [name-substring]
Unbundles the gfx950 code object from the .hip_fatbin section and reads its AMDGPU metadata note with llvm-readelf."""
import re, struct, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else 'open_l2o_amd/libl2o_hip.so'
pat = sys.argv[2] if len(sys.argv) > 2 else ''
data = open(lib, 'rb').read()
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
start = 0
while True:                                     # one bundle per translation unit (csrc/Makefile links two)
    i = data.find(MAGIC, start)
    if i < 0:
        break
    start = i + len(MAGIC)
    n, = struct.unpack_from('<Q', data, i + 24)
    pos = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from('<QQQ', data, pos)
        triple = data[pos + 24:pos + 24 + tl].decode()
        pos += 24 + tl
        if 'gfx950' in triple:
            with tempfile.NamedTemporaryFile(suffix='.co') as f:
                f.write(data[i + off:i + off + size]); f.flush()
                out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
            for b in out.split('  - .agpr_count:')[1:]:
                name = re.search(r'\.name:\s+(\S+)', b).group(1)
                if pat not in name:
                    continue
                g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1))
                print('%-70s vgpr %3d (agpr %3d) sgpr %3d (spilled %3d) spill %3d scratch %4d lds %6d' % (
                    name[:70], g('vgpr_count'), int(b.split('\n')[0]), g('sgpr_count'), g('sgpr_spill_count'), g('vgpr_spill_count'),
                    g('private_segment_fixed_size'), g('group_segment_fixed_size')))
