"""VGPR / AGPR / spill / LDS figures of the kernels inside libvisfly_amd.so (the gfx950 code objects of the offload bundles, llvm-readelf
--notes).  python tools/kernel_resources.py [regex] [shared object]"""
import os, re, shutil, struct, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "visfly_amd", "csrc", "libvisfly_amd.so"), "rb").read()
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
rows, pos, tmp = [], so.find(MAGIC), tempfile.mkdtemp()
while pos != -1:
    p = pos + len(MAGIC)
    (num,), p = struct.unpack_from("<Q", so, p), p + 8
    for _ in range(num):
        off, size, tl = struct.unpack_from("<QQQ", so, p)
        p += 24
        triple, p = so[p:p + tl].decode(), p + tl
        if "gfx950" in triple and size:
            f = os.path.join(tmp, "co")
            open(f, "wb").write(so[pos + off:pos + off + size])
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            for blk in out.split("- .agpr_count")[1:]:
                g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                rows.append((g("vgpr_count"), int(re.search(r"^:\s+(\d+)", blk).group(1)), g("vgpr_spill_count"),
                             g("group_segment_fixed_size"), re.search(r"\.name:\s+(\S+)", blk).group(1)))
    pos = so.find(MAGIC, pos + len(MAGIC))
shutil.rmtree(tmp)
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
names = [r[4] for r in rows]
if filt:
    names = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print(f"{'vgpr':>5} {'agpr':>5} {'spill':>5} {'lds':>7}  kernel")
for (v, a, sp, lds, _), d in sorted(zip(rows, names), reverse=True):
    if pat.search(d):
        print(f"{v:5d} {a:5d} {sp:5d} {lds:7d}  {d[:160]}")
