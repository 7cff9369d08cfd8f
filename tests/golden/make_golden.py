#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE ITSELF (build container only).

Needs oracle/_ref/ (built by `make -C oracle ref` from /root/reference, see oracle/Makefile):
  libref_shim.so  reference headers (tables, encoder, partitions, PseudoDBdata, kseqpp) as they lie
  libref_dp.so    the reference's own scalar DP checker (cudasw4.cuh:2331-2392), sliced at build time
  makedb          the reference's makedb binary

Outputs (data only — inputs and expected outputs):
  allqueries.fasta                 the reference's query set (data file, runpeakbenchmark.sh:26)
  allqueries_db/aq*                dbdata files written by the reference makedb for that FASTA
  ref_tables.json                  BLOSUM 21x21 and 25x25 tables, partition bounds, encoder map, pseudo-DB residues
  ref_scores.json                  scores computed by the reference DP:
                                     pseudo[L][q]      20 queries x pseudo subject of length L (seed 42)
                                     allvsall[i][j]    20 x 20
                                     pairs[]           seeded random pairs incl. code-20 letters, empty-ish and ragged lengths
                                     long_subject      one > 8000-residue subject (partition 35) vs all queries
"""
import ctypes, json, os, shutil, subprocess, random
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.abspath(os.path.join(here, "..", ".."))
refdir = os.path.join(root, "oracle", "_ref")
REF = os.environ.get("SW_REFERENCE_DIR", "/root/reference")

shim = ctypes.CDLL(os.path.join(refdir, "libref_shim.so"))
dp = ctypes.CDLL(os.path.join(refdir, "libref_dp.so"))
shim.ref_fasta_open.restype = ctypes.c_void_p
shim.ref_fasta_seq.restype = ctypes.c_char_p
shim.ref_fasta_header.restype = ctypes.c_char_p
shim.ref_pseudodb.restype = ctypes.c_size_t
for f in (shim.ref_fasta_count, shim.ref_fasta_seqlen, shim.ref_fasta_seq, shim.ref_fasta_header, shim.ref_fasta_close):
    f.argtypes = [ctypes.c_void_p] + ([ctypes.c_int] if f not in (shim.ref_fasta_count, shim.ref_fasta_close) else [])
dp.ref_dp_score_converted.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]

def ref_encode(s: bytes) -> bytes:
    out = (ctypes.c_int8 * len(s))()
    shim.ref_encode(s, out, ctypes.c_size_t(len(s)))
    return bytes(bytearray(np.frombuffer(out, dtype=np.int8).astype(np.uint8)))

def ref_score(a: bytes, b: bytes, gop=-11, gex=-1) -> int:
    return int(dp.ref_dp_score_converted(a, b, len(a), len(b), gop, gex))

def pseudo_codes(L, seed=42) -> bytes:
    lr = (L + 3) // 4 * 4
    chars = (ctypes.c_int8 * lr)()
    lens = (ctypes.c_int32 * 1)()
    offs = (ctypes.c_uint64 * 2)()
    shim.ref_pseudodb(ctypes.c_size_t(1), ctypes.c_int32(L), ctypes.c_int(seed), chars, lens, offs)
    return bytes(bytearray(np.frombuffer(chars, dtype=np.int8)[:L].astype(np.uint8)))

def main():
    fasta = os.path.join(REF, "allqueries.fasta")
    shutil.copyfile(fasta, os.path.join(here, "allqueries.fasta"))

    h = shim.ref_fasta_open(fasta.encode())
    nq = shim.ref_fasta_count(h)
    seqs = [shim.ref_fasta_seq(h, i) for i in range(nq)]
    headers = [shim.ref_fasta_header(h, i).decode() for i in range(nq)]
    shim.ref_fasta_close(h)
    enc = [ref_encode(s) for s in seqs]

    # ---- tables
    tables = {}
    for which in (45, 50, 62, 80):
        buf = (ctypes.c_int8 * 441)()
        assert shim.ref_blosum21(which, buf) == 21
        tables[str(which)] = [int(x) for x in buf]
    tables25 = {}
    for which in (45, 50, 62, 80):  # types.hpp:205-396, letter order ARNDCQEGHILKMFPSTWYVBJZX*
        buf = (ctypes.c_int8 * 625)()
        assert shim.ref_blosum25(which, buf) == 25
        tables25[str(which)] = [int(x) for x in buf]
    bounds = (ctypes.c_int32 * 64)()
    nb = shim.ref_partition_boundaries(bounds, 64)
    encmap = list(ref_encode(bytes(range(256))))
    out_tables = {
        "blosum21": tables,
        "blosum25": tables25,
        "partition_boundaries": [int(bounds[i]) for i in range(nb)],
        "encode_map_256": encmap,
        "pseudodb_seed42_first2048": list(pseudo_codes(2048, 42)),
        "pseudodb_seed7_first64": list(pseudo_codes(64, 7)),
        "query_lengths": [len(s) for s in seqs],
        "query_headers": headers,
    }
    with open(os.path.join(here, "ref_tables.json"), "w") as f:
        json.dump(out_tables, f, separators=(",", ":"))

    # ---- scores by the reference DP
    scores = {"gop": -11, "gex": -1, "matrix": 62}
    scores["pseudo"] = {str(L): [ref_score(q, pseudo_codes(L)) for q in enc] for L in (128, 256, 512, 768, 1024, 2048)}
    scores["allvsall"] = [[ref_score(enc[i], enc[j]) for j in range(nq)] for i in range(nq)]

    rng = random.Random(20241001)
    pairs = []
    def rnd_seq(n, p20=0.0):
        return bytes(20 if rng.random() < p20 else rng.randrange(20) for _ in range(n))
    shapes = [(1, 1), (1, 7), (7, 1), (2, 33), (16, 16), (17, 15), (31, 300), (300, 31), (64, 64), (65, 63),
              (127, 129), (128, 128), (129, 513), (255, 47), (256, 48), (257, 49), (511, 80), (512, 512),
              (513, 100), (700, 1281), (33, 8001)]
    for (a, b) in shapes:
        for p20 in (0.0, 0.1):
            qa, sb = rnd_seq(a, p20), rnd_seq(b, p20)
            pairs.append({"q": list(qa), "s": list(sb), "score": ref_score(qa, sb)})
    # homologous pairs (mutated copies) so that scores are large / gaps are used
    for n in (50, 200, 600, 1500):
        base = rnd_seq(n)
        mut = bytearray()
        for c in base:
            r = rng.random()
            if r < 0.05: continue                      # deletion
            if r < 0.10: mut.append(rng.randrange(20)) # insertion
            mut.append(c if rng.random() > 0.15 else rng.randrange(20))
        pairs.append({"q": list(base), "s": list(bytes(mut)), "score": ref_score(base, bytes(mut))})
    # other gap penalties (the DP takes them as arguments even though the reference CLI never forwards them)
    for (gop, gex) in ((-5, -2), (-13, -3), (-3, -3)):
        qa, sb = rnd_seq(90), rnd_seq(110)
        pairs.append({"q": list(qa), "s": list(sb), "gop": gop, "gex": gex, "score": ref_score(qa, sb, gop, gex)})
    scores["pairs"] = pairs

    long_subject = b"".join(enc[10:14])                 # 1500+2005+2504+3005 = 9014 > 8000 -> partition 35
    scores["long_subject"] = {"concat_of_queries": [10, 11, 12, 13], "length": len(long_subject),
                              "scores": [ref_score(q, long_subject) for q in enc]}
    with open(os.path.join(here, "ref_scores.json"), "w") as f:
        json.dump(scores, f, separators=(",", ":"))

    # ---- dbdata files written by the reference makedb
    dbdir = os.path.join(here, "allqueries_db")
    shutil.rmtree(dbdir, ignore_errors=True)
    os.makedirs(dbdir)
    subprocess.check_call([os.path.join(refdir, "makedb"), fasta, os.path.join(dbdir, "aq")], stdout=subprocess.DEVNULL)
    print("golden fixtures written:", sorted(os.listdir(here)))

if __name__ == "__main__":
    main()
