"""Host-side restatement of the index maps behind "the draws written once" (csrc/kernels_fullrank_batch.hip, k_fb_vjp): the LDS-DMA lane ->
chunk map that gathers a VJP B piece from the draws' product-orientation planes, and the per-lane addresses of the transposing reads
(ds_read_b64_tr_b16; semantics probed on the device by tools/ubench/tr_b16_probe.hip: lane c of a 16-lane group receives, as element j, element
c % 4 of the 8-byte chunk whose address lane 4 j + c / 4 of the group supplies).  Checked here without a GPU: the fragment every lane ends up
with is the MFMA B fragment of eps' (dim n = lane % 32, h = lane / 32: slots e <-> samples 8 (e / 4) + 4 h + e % 4 of the 16-sample group), and
every 32-lane read group touches 32 distinct 8-byte bank pairs (no LDS bank conflict).  The GPU tests check the numbers; this pins the layout."""
import numpy as np


def product_plane_chunk(eps, kg, s, hp):
    """The 16-byte chunk (eight f16 slots) lane (sample s, h = hp) holds in fragment (mb32, kg) of the product-orientation planes:
    dims 16 kg + 8 (e / 4) + 4 hp + e % 4 of sample s (fr_planes.h).  eps: [dim][sample]."""
    return np.array([eps[16 * kg + 8 * (e // 4) + 4 * hp + e % 4, s] for e in range(8)])


def dma_piece(eps, jb, mg):
    """LDS image (64 chunks of eight values) of the B piece of dim block jb (32 dims), sample group mg (16 samples): DMA lane p fetches chunk
    (kgpar, hp, sl) with m8 = p >> 3, s8 = m8 >> 2, hp = (m8 >> 1) & 1, kgpar = m8 & 1, sl = 8 s8 + ((p & 7) ^ 4 hp)."""
    img = np.zeros((64, 8))
    lines = []
    for p in range(64):
        m8 = p >> 3
        s8, hp, kgpar = m8 >> 2, (m8 >> 1) & 1, m8 & 1
        sl = 8 * s8 + ((p & 7) ^ (4 * hp))
        s = 16 * mg + sl                       # sample (global): fragment mb32 = s // 32, lane s % 32 + 32 hp
        img[p] = product_plane_chunk(eps, 2 * jb + kgpar, s, hp)
        lines.append((2 * jb + kgpar, s // 32, ((s % 32) + 32 * hp) // 8))   # (fragment, 128-byte line inside its plane)
    # eight consecutive DMA lanes fetch one 128-byte line
    for q in range(8):
        assert len(set(lines[8 * q:8 * q + 8])) == 1
    return img


def tr_read(img_bytes_as_halfs, addr_of_lane):
    """ds_read_b64_tr_b16: img as a flat array of f16 slots (2 bytes each); addr_of_lane[lane] in bytes.  Returns [64][4]."""
    out = np.zeros((64, 4))
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        for j in range(4):
            src = 16 * g + 4 * j + c // 4
            out[lane, j] = img_bytes_as_halfs[addr_of_lane[src] // 2 + c % 4]
    return out


def lane_read_addresses():
    addr = np.zeros(64, dtype=np.int64)
    for lane in range(64):
        G4, c16 = lane >> 4, lane & 15
        kgpar, hq, hp, half, sl = G4 & 1, G4 >> 1, c16 & 1, (c16 >> 1) & 1, 4 * (G4 >> 1) + (c16 >> 2)
        addr[lane] = 16 * (8 * (2 * hp + kgpar) + ((sl & 7) ^ (4 * hp))) + 8 * half
    return addr


def test_transposed_reads_assemble_the_vjp_b_fragment():
    rng = np.random.default_rng(3)
    d, M = 128, 64
    eps = rng.standard_normal((d, M))
    addr = lane_read_addresses()
    for jb in range(d // 32):
        for mg in range(M // 16):
            flat = dma_piece(eps, jb, mg).reshape(-1)          # 64 chunks x 8 slots, 2 bytes per slot
            lo = tr_read(flat, addr)                            # slots e = 0 .. 3
            hi = tr_read(flat, addr + 512)                      # slots e = 4 .. 7 (the second read: + 512 bytes)
            for lane in range(64):
                n, h = lane % 32, lane // 32
                for e in range(8):
                    want = eps[32 * jb + n, 16 * mg + 8 * (e // 4) + 4 * h + e % 4]
                    got = (lo if e < 4 else hi)[lane, e % 4]
                    assert got == want, (jb, mg, lane, e)


def test_every_read_group_is_bank_conflict_free():
    addr = lane_read_addresses()
    for off in (0, 512):
        for half_wave in (range(0, 32), range(32, 64)):         # ds_read_b64_tr_b16 is served in two groups of 32 lanes
            pairs = {((addr[l] + off) // 8) % 32 for l in half_wave}   # 64 banks of 4 bytes = 32 pairs of 8 bytes
            assert len(pairs) == 32
