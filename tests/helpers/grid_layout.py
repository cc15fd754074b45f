"""Python mirror of pfg::grid_layout (csrc/pfg_grid_kernel.hpp): where the whole-GPU window keeps what inside its
per-window scratch.  TEST INFRASTRUCTURE (tests read the REPLAY CDF back from the scratch); the size is asserted
against pfg_scratch_bytes."""
REC = {("svm", "f64"): 4, ("garch", "f64"): 6, ("lgssm", "f64"): 6, ("svm", "f32"): 4, ("garch", "f32"): 8, ("lgssm", "f32"): 8}


def _al(x):
    return (x + 255) & ~255


def grid_layout(model, dtype, N, replay):
    rs = 8 if dtype == "f64" else 4
    L = {"N": N}
    L["TILE"] = 1024 if N <= (1 << 19) else 2048       # 256 threads x 4 | 8 children
    L["NT"] = 256
    L["G"] = (N + L["TILE"] - 1) // L["TILE"]
    S = 64
    while (N + S - 1) // S > 16384:
        S <<= 1
    L["S"], L["C"] = S, (N + S - 1) // S
    o = 0
    L["lw"] = []
    for _ in range(2):
        L["lw"].append(o); o = _al(o + N * rs)
    L["rec"] = []
    for _ in range(2):
        L["rec"].append(o); o = _al(o + N * REC[(model, dtype)] * rs)
    L["part"] = []
    for _ in range(2):
        L["part"].append(o); o = _al(o + (7 * L["G"] + 8) * 8)
    L["rng"] = o; o = _al(o + L["G"] * L["NT"] * 16)
    L["head"] = o; o = _al(o + 32 * 8)
    if not replay:
        L["cs"] = []
        for _ in range(2):
            L["cs"].append(o); o = _al(o + N * 8)
        L["tab"] = o; o = _al(o + (128 + 256) * 8)
        L["consts"] = o; o = _al(o + 23 * 8)
    if replay:
        L["cdf"] = o; o = _al(o + N * 8)
        L["coarse"] = o; o = _al(o + L["C"] * 8)
        L["walk_i"] = o; o = _al(o + N * 4)
        L["walk_p"] = o; o = _al(o + N * 8)
        L["walk_q"] = o; o = _al(o + N * 8)
        L["walk_s"] = o; o = _al(o + N * 8)
        L["cdfx"] = o; o = _al(o + (512 + 6 * 1024 + 8) * 8)
    L["bytes"] = o
    return L
