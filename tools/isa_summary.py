"""Per-kernel resource summary of a gfx950 assembly file (hipcc -S --cuda-device-only): registers, spills, LDS, MFMA / DMA counts.
Usage: python tools/isa_summary.py file.s [...]"""
import re
import sys

VM0 = r'vmcnt\(0\)'


def summarize(path):
    txt = open(path).read()
    # metadata block: one YAML entry per kernel
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        agpr, lds, name, scratch, sgpr, vgpr, spill = m.groups()
        body = re.search(r"^%s:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), txt, re.S | re.M)
        b = body.group(1) if body else ""
        cnt = lambda pat: len(re.findall(pat, b))
        print(f"{name[:70]:70s} vgpr {vgpr:>3} (agpr {agpr:>3}) sgpr {sgpr:>3} spill {spill:>3} scratch {scratch:>4} lds {lds:>6} | mfma {cnt(r'v_mfma'):5d} "
              f"dma {cnt(r'global_load_lds|buffer_load.*lds'):3d} gload {cnt(r'global_load_dword'):4d} gstore {cnt(r'global_store|buffer_store'):4d} "
              f"ds_r {cnt(r'ds_read'):4d} ds_w {cnt(r'ds_write'):4d} vmcnt0 {cnt(VM0):3d} barrier {cnt(r's_barrier'):2d}")


for p in sys.argv[1:]:
    summarize(p)
