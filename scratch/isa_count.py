"""Static inventory of the vector instructions of one a-trous kernel instance in a device assembly listing (hipcc -S --cuda-device-only):
   python scratch/isa_count.py <listing.s> [STEP] [R]"""
import collections, sys
path = sys.argv[1]
step, rows = (sys.argv[2] if len(sys.argv) > 2 else "1"), (sys.argv[3] if len(sys.argv) > 3 else "4")
name = f"_ZN3vhr23svgf_atrous_tile_kernelILi{step}ELi{rows}EEE"
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(name)][0]
end = [i for i, l in enumerate(lines[start:]) if l.strip().startswith("s_endpgm")][0] + start
body = lines[start:end]
isv = lambda l: l.strip().startswith("v_")
bar = [i for i, l in enumerate(body) if "s_barrier" in l]
logs = [i for i, l in enumerate(body) if "v_log_f32" in l]
exps = [i for i, l in enumerate(body) if "v_exp_f32" in l]
regions = {"pre-barrier (static, both branches)": (0, bar[0]), "barrier -> first log": (bar[0], logs[0]), "taps": (logs[0], exps[-1] + 1), "epilogue": (exps[-1] + 1, len(body))}
for nm, (a, b) in regions.items():
    c = collections.Counter(l.strip().split()[0] for l in body[a:b] if isv(l))
    lds = sum(1 for l in body[a:b] if l.strip().startswith("ds_"))
    print(f"{nm}: {sum(c.values())} vector, {lds} LDS, {sum(1 for l in body[a:b] if 's_nop' in l)} s_nop, {sum(1 for l in body[a:b] if 's_waitcnt' in l)} s_waitcnt")
    print("   ", c.most_common(16))
for l in lines[end:end + 80]:
    if any(k in l for k in ("NumVgprs", "Occupancy", "LDSByteSize", "ScratchSize")):
        print(l.strip())
