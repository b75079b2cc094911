"""profiles/r02_static_kernel_facts.md from the resource-usage remarks and the ISA (see static_kernel_facts.sh)."""
import re, collections, subprocess
import sys
work=sys.argv[1]; repo=sys.argv[2]
ru=open(work+'/ru.txt').read()
rows=[]
blocks=ru.split("Function Name: ")[1:]
def field(b, key):
    m=re.search(re.escape(key)+r":? *(\d+)", b)
    return m.group(1) if m else "?"
def dem(n):
    return subprocess.run(['/usr/bin/c++filt', n],capture_output=True,text=True).stdout.strip().split('(')[0].replace('void ','').replace('rspt::','')
want=['k_raygen','k_trace_w4<false, 0, false, false>','k_trace_w4<true, 0, false, false>','k_trace_w4<false, 0, true, false>','k_trace_w4<true, 0, true, false>','k_trace_w4<false, 0, false, true>','k_trace_w4<true, 0, false, true>','k_trace_fixup<false, 0, false, false>','k_texture','k_shade','k_bin_count','k_bin_scatter','k_film','k_ao_spawn','k_ao_resolve','k_dl_hit','k_dl_nee','k_vol_shade','k_vol_tr','k_tile_serial<false, false>','k_trace_pw<false, 0>','k_trace<false, 0, false, false, false>','k_trace<false, 0, false, true, false>']
out=["# Static facts about the final kernels (hipcc -Rpass-analysis=kernel-resource-usage and the gfx950 ISA of librspt.hip)","",
"Generated on the build host from the committed sources (no GPU needed) by `tools/static_kernel_facts.sh`.","",
"| kernel | VGPRs | scratch B/lane | LDS B/workgroup | waves/SIMD | VGPR spills |","|---|---|---|---|---|---|"]
seen=set()
for b in blocks:
    n=dem(b.split()[0])
    if n in want and n not in seen:
        seen.add(n)
        out.append("| `%s` | %s | %s | %s | %s | %s |" % (n, field(b," VGPRs"), field(b,"ScratchSize [bytes/lane]"), field(b,"LDS Size [bytes/block]"), field(b,"Occupancy [waves/SIMD]"), field(b,"VGPRs Spill")))
asm=open(work+'/final.s').read()
def body(prefix):
    m=re.search(r"^(%s\S*):.*?s_endpgm" % re.escape(prefix), asm, re.S|re.M)
    return m.group(0)
def hist(b, pat):
    c=collections.Counter(re.findall(r"^\s+(%s\w*)" % pat, b, re.M))
    return ", ".join("%s x%d" % (k,v) for k,v in sorted(c.items(), key=lambda kv:-kv[1]))
def count(b, k):
    return len(re.findall(r"^\s+"+k, b, re.M))
for name,prefix in [("k_trace_w4<closest, queue mode, no instances, no alpha masks>","_ZN4rspt10k_trace_w4ILb0ELi0ELb0ELb0E"),("k_shade","_ZN4rspt7k_shadeE"),("k_texture","_ZN4rspt9k_textureE")]:
    b=body(prefix)
    n_ins=len(re.findall(r"^\s+[a-z]\w+", b, re.M))
    out += ["", "## `%s`: %d instructions (static)" % (name, n_ins), "", "memory instructions: "+hist(b, r"(?:global|flat|ds|scratch|buffer)_"), ""]
    if "shade" in name:
        ks=["v_div_scale_f32","v_div_fmas_f32","v_div_fixup_f32","v_rcp_f32","v_sqrt_f32","v_fma_f64","v_mul_f64","v_mad_u64_u32","v_readlane_b32","v_writelane_b32","s_nop"]
        out += ["arithmetic that stands out: "+", ".join("%s x%d" % (k, count(b,k)) for k in ks), ""]
out += ["", "Reading: the traversal step of `k_trace_w4` is the block with the seven `global_load_dwordx4` of a record (or the seven",
"`ds_read_b128` of an LDS-resident root-side record), the stack pop is the `ds_read_b64` (+ a conditional `global_load_dwordx2` from",
"the spill rows), and there is no flat instruction in the kernel (DESIGN.md §5). `k_shade`'s correctly rounded divisions",
"(`v_div_scale/fmas/fixup` + `v_rcp`) and square roots are what bit-exact parity with Rust's IEEE `/` and `sqrt` costs; its",
"register count (2 waves/SIMD) is the first open end of DESIGN.md §9. Template arguments of `k_trace_w4`: <any-hit, output mode, object instances, alpha masks>; the `true` variants are only launched for scenes with instances (SURVEY 8(f) #2) / alpha-masked meshes."]
open(repo+'/profiles/r02_static_kernel_facts.md','w').write("\n".join(out)+"\n")
print("\n".join(out))
