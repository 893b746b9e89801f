"""hisparse_amd.datasets — the benchmark matrices of the reference, as seeded synthetic stand-ins.

The reference's datasets (sw/bm.sh:3-17, paper Table 2) live in a Google-Drive archive that is not part
of the checkout (datasets/download.sh; git-lfs blob missing) and there is no network here, so every
named configuration below can be produced two ways:
  * from the real `.npz` if the caller has it (`path=`), through the same loader the reference uses
    (load_csr_matrix_from_float_npz, sw/data_loader.h:51-70);
  * otherwise from a seeded generator of the same shape / non-zero count.
Each entry also carries the numeric mode and bank sizes BASELINE.json's `configs` pair it with.
"""
from dataclasses import dataclass

from . import host


@dataclass(frozen=True)
class Config:
    name: str
    impl: str            # fixed | float_pob | float_stall
    rows: int
    cols: int
    kind: str            # generator kind (host.CSRMatrix.generate)
    a: float             # powerlaw: target nnz;  uniform: nnz per row
    b: float             # powerlaw: skew exponent; bernoulli: density
    c: float             # value scale
    seed: int
    npz: str             # file name of the real dataset (sw/bm.sh:4-17)
    skip_empty_rows: bool = True
    note: str = ""


# value scale: the reference overwrites every value with `1 / num_cols` (integer division => 0.0,
# sw/benchmark.cpp:411); the stand-ins use non-degenerate values instead so that parity means something.
CONFIGS = {
    # BASELINE.json configs[0]: plumbing case
    "csim_1k": Config("csim_1k", "fixed", 1000, 1000, "bernoulli", 0, 0.01, 0.5, 1, "", True,
                      "1k x 1k, 1 % dense (values |N(0,0.5)| clipped by the unsigned fixed-point type)"),
    # configs[1]: the north-star workload
    "ogbl_ppa": Config("ogbl_ppa", "fixed", 576289, 576289, "powerlaw", 42463862, 0.35, 1.0, 42,
                       "ogbl_ppa_576K_42M_csr_float32.npz"),
    # configs[2]
    "transformer_50": Config("transformer_50", "float_pob", 512, 33288, "bernoulli", 0, 0.5, 0.05, 50,
                             "transformer_50_512_33288_csr_float32.npz"),
    # configs[3]
    "ogbn_products": Config("ogbn_products", "float_stall", 2449029, 2449029, "powerlaw", 123718280, 0.43, 1.0, 43,
                            "ogbn_products_2M_124M_csr_float32.npz"),
    # configs[4]
    "mouse_gene": Config("mouse_gene", "fixed", 45101, 45101, "powerlaw", 28967291, 0.30, 0.1, 44,
                         "mouse_gene_45K_29M_csr_float32.npz"),
    # SURVEY.md section 8d's own recipe for the two big graphs: symmetric R-MAT (a=.57 b=.19 c=.19), seeds 42 / 43 -- a second
    # stand-in with a much harder degree distribution (hub rows of 10^5 non-zeros) than the Chung-Lu matrices above
    # (`a` counts the entries DRAWN; R-MAT repeats itself a lot at this density -- 62.2 M draws leave ~42.4 M distinct entries)
    "ogbl_ppa_rmat": Config("ogbl_ppa_rmat", "fixed", 576289, 576289, "rmat", 62.2e6, 1.0, 1.0, 42, ""),
    "ogbn_products_rmat": Config("ogbn_products_rmat", "float_stall", 2449029, 2449029, "rmat", 161e6, 1.0, 1.0, 43, ""),
    # the rest of the reference's benchmark list (sw/bm.sh:3-17; shapes and non-zero counts from the file names and paper Table 2;
    # skew exponents: guesses by graph family -- social graphs with heavy hubs steeper than the co-star / protein graphs)
    "gplus": Config("gplus", "fixed", 107614, 107614, "powerlaw", 13673453, 0.45, 1.0, 45, "gplus_108K_13M_csr_float32.npz"),
    "hollywood": Config("hollywood", "fixed", 1069126, 1069126, "powerlaw", 112751422, 0.35, 1.0, 46, "hollywood_1M_113M_csr_float32.npz"),
    "pokec": Config("pokec", "fixed", 1632803, 1632803, "powerlaw", 30622564, 0.30, 1.0, 47, "pokec_1633K_31M_csr_float32.npz"),
    "transformer_60": Config("transformer_60", "float_pob", 512, 33288, "bernoulli", 0, 0.4, 0.05, 60, "transformer_60_512_33288_csr_float32.npz"),
    "transformer_70": Config("transformer_70", "float_pob", 512, 33288, "bernoulli", 0, 0.3, 0.05, 70, "transformer_70_512_33288_csr_float32.npz"),
    "transformer_80": Config("transformer_80", "float_pob", 512, 33288, "bernoulli", 0, 0.2, 0.05, 80, "transformer_80_512_33288_csr_float32.npz"),
    "transformer_90": Config("transformer_90", "float_pob", 512, 33288, "bernoulli", 0, 0.1, 0.05, 90, "transformer_90_512_33288_csr_float32.npz"),
    "transformer_95": Config("transformer_95", "float_pob", 512, 33288, "bernoulli", 0, 0.05, 0.05, 95, "transformer_95_512_33288_csr_float32.npz"),
    # one rank's share of mouse_gene split 8 / 4 ways by non-zeros (bench.py --gpus 8 / 4): the small-slab regime (tools/, tests)
    "mouse_gene_slab8": Config("mouse_gene_slab8", "fixed", 5638, 45101, "powerlaw", 3620911, 0.30, 0.1, 44, ""),
    "mouse_gene_slab2": Config("mouse_gene_slab2", "fixed", 22550, 45101, "powerlaw", 14483645, 0.30, 0.1, 44, ""),
    "mouse_gene_slab4": Config("mouse_gene_slab4", "fixed", 11275, 45101, "powerlaw", 7241822, 0.30, 0.1, 44, ""),
    # ogbn-products / pokec with the same rows and non-zeros over 1/2 and 1/4 of the columns: twice / four times the non-zeros per (row range,
    # x sub-tile) unit -- what heavier units would buy the OWNER path (tools/history/r03/heavy_units.sh; experiments only)
    "ogbn_half_cols": Config("ogbn_half_cols", "float_stall", 2449029, 1224514, "powerlaw", 123718280, 0.43, 1.0, 43, ""),
    "ogbn_quarter_cols": Config("ogbn_quarter_cols", "float_stall", 2449029, 612257, "powerlaw", 123718280, 0.43, 1.0, 43, ""),
    "pokec_quarter_cols": Config("pokec_quarter_cols", "fixed", 1632803, 408200, "powerlaw", 30622564, 0.30, 1.0, 47, ""),
    # small relatives for tests / smoke
    "ppa_small": Config("ppa_small", "fixed", 40000, 70000, "powerlaw", 1400000, 0.35, 1.0, 7, ""),
    "nn_small": Config("nn_small", "float_pob", 512, 33288, "bernoulli", 0, 0.05, 0.05, 95, ""),
}


# sw/bm.sh's sweep, in its order, with the fixed-point throughput the paper reports for each on the U280 (GOPS, Table 3) -- the
# reference's own bar per matrix.  bm.sh runs every matrix in the numeric mode of the bitstream it is given; Table 3 is the fixed one.
BM_LIST = [("gplus", 21.2), ("ogbl_ppa", 24.4), ("hollywood", 24.9), ("pokec", 11.2), ("ogbn_products", 20.6), ("mouse_gene", 27.2),
           ("transformer_50", 21.9), ("transformer_60", 18.9), ("transformer_70", 16.5), ("transformer_80", 14.8), ("transformer_90", 9.7),
           ("transformer_95", 5.7)]


# The paper's Table 7: the matrices it quotes in all three numeric modes -- GOPS on the U280 as (fixed, float_pob "PB", float_stall "RI").
# sw/bm.sh:19-35 runs the whole list in the mode of the bitstream it is given (ob = 1 for float_pob, else 8).
BM_FLOAT = [("transformer_80", 14.8, 13.4, 6.3), ("mouse_gene", 27.2, 25.0, 13.1), ("pokec", 11.2, 3.4, 9.1), ("ogbn_products", 20.6, 6.7, 16.3)]


def load(name, path=None, scale=1.0):
    """(Config, CSRMatrix).  `scale` < 1 shrinks rows, cols and nnz of a generated stand-in proportionally."""
    cfg = CONFIGS[name]
    if path:
        return cfg, host.load_csr_matrix_from_float_npz(path)
    rows, cols, a = cfg.rows, cfg.cols, cfg.a
    if scale != 1.0:
        rows, cols = max(128, int(rows * scale)), max(8, int(cols * scale))
        if cfg.kind == "powerlaw":
            a = max(1.0, a * scale)
    return cfg, host.CSRMatrix.generate(cfg.kind, rows, cols, a=a, b=cfg.b, c=cfg.c, seed=cfg.seed)
