"""CPU: the hand-written LZ4 blocks of corpus.synthetic_blocks (copy chains, straddling and periodic sources: what the GPU
decoders' relinking and cooperative copies are tested with) are what their generator says they are: the oracle (lz4_flex's
decoder restated) and liblz4 decode them to the generator's plain text."""
import corpus
import oracle_api as O


def test_the_oracle_decodes_the_synthetic_blocks_to_their_plain_text():
    blocks = corpus.synthetic_blocks(sizes=(150000,) * 4 + (5000,) * 8 + (1 << 20,))
    for comp, plain in blocks:
        assert O.decompress(comp, len(plain)) == ("ok", plain)
        assert O.c_decompress(comp, len(plain)) == plain


def test_the_pcd_model_decodes_them_in_every_geometry():
    import pcd_model as M
    from test_pcd_model import geometries
    blocks = corpus.synthetic_blocks(seed=5, sizes=(60000,) * 3 + (3000,) * 6)
    for comp, plain in blocks:
        for _name, params in geometries():
            got, _st = M.decode(comp, len(plain), params)
            assert got == plain
