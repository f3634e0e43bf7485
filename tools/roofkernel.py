"""Which kernel of a rocprofv3 trace is the roofline kernel of the cached EmbeddingBag gather, and what it moves.

Since round 4's second session the step of the Criteo layout with the dot interaction has no stand-alone gather: the cache
rows are the operand loads of the interaction forward (`k_interact_fwd_s<D/4, slabs, true>`, cdlrm_gather_interact_fwd).  Traces
of other paths (multi-hot bags, "cat", `fuse_gather = False`) still hold `k_embbag_fwd_arange_p`."""

GATHER = "k_embbag_fwd_arange"
FUSED = "k_interact_fwd_s<"


def kind(kernel_name: str):
    n = kernel_name.replace("void ", "").split("(")[0].strip()
    if GATHER in n:
        return "gather"
    if FUSED in n and n.endswith("true>"):
        return "fused"
    return None


def pick(kernel_names):
    """'fused' when the trace holds the fused kernel, else 'gather' (None: neither)."""
    kinds = {kind(n) for n in kernel_names}
    return "fused" if "fused" in kinds else ("gather" if "gather" in kinds else None)


def bytes_per_launch(which: str, B: int, T: int, D: int):
    """(survey_basis, kernel_basis): SURVEY 8(d) prices a lookup at 8D + 16 (row in, pooled row out, int64 index + offset);
    the stand-alone gather itself moves 8D + 4 (int32 slot id, arange offsets are not read); the fused kernel reads the row and
    the slot id (4D + 4 per lookup), the dense feature (4D per sample) and writes the interaction row (D + T(T+1)/2 floats,
    padded to a multiple of 4) -- the pooled rows are never written."""
    look = B * T
    survey = look * (8 * D + 16)
    if which == "gather":
        return survey, look * (8 * D + 4)
    width = (D + T * (T + 1) // 2 + 3) // 4 * 4
    return survey, look * (4 * D + 4) + B * 4 * D + B * 4 * width
