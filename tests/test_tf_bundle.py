"""Checkpoint format (ganmf_amd/tf_bundle.py) against the reference's surviving tf.train.Saver files
(tests/golden/*.index are those files, copied by oracle/make_golden.py which also asserts, where the 2 MB
.data file is available, that the writer reproduces both files byte for byte).  CPU only."""
import json
import os

import numpy as np
import pytest

from ganmf_amd import tf_bundle as tb


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    assert tb.crc32c(b"") == 0
    assert tb.crc32c(bytes(32)) == 0x8A9136AA              # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43     # RFC 3720 B.4: 32 bytes of ones
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E       # RFC 3720 B.4: incrementing
    a = np.arange(1000, dtype=np.float32)
    whole = tb.crc32c(a)
    assert tb.crc32c(a[500:], tb.crc32c(a[:500])) == whole  # streaming form


def test_reads_reference_index_and_checks_tensor_crcs(golden_dir):
    header, e = tb.read_index(os.path.join(golden_dir, "kat1_GANMF_item.index"))
    # SURVEY Appendix C
    want = {"autoencoder/decoding/bias": ((1884,), 0), "autoencoder/decoding/kernel": ((133, 1884), 7536),
            "autoencoder/encoding/bias": ((133,), 1009824), "autoencoder/encoding/kernel": ((1884, 133), 1010356),
            "generator/item_embeddings": ((1884, 1), 2012644), "generator/user_embeddings": ((17632, 1), 2020180)}
    assert {k: (v["shape"], v["offset"]) for k, v in e.items()} == want
    assert all(v["dtype"] == 1 and v["size"] == 4 * int(np.prod(v["shape"])) for v in e.values())
    # the tensors of that checkpoint held as fixtures must hash to the CRCs TensorFlow stored
    t = np.load(os.path.join(golden_dir, "kat1_checkpoint_tensors.npz"))
    for key, name in (("U", "generator/user_embeddings"), ("V", "generator/item_embeddings"),
                      ("be", "autoencoder/encoding/bias"), ("bd", "autoencoder/decoding/bias")):
        assert tb._mask(tb.crc32c(np.ascontiguousarray(t[key], dtype="<f4"))) == e[name]["crc32c"], name


def test_reads_second_reference_index(golden_dir):
    chk = json.load(open(os.path.join(golden_dir, "kat1_bundle_check.json")))
    assert chk["writer_reproduces_reference_bytes"] is True
    _, e = tb.read_index(os.path.join(golden_dir, "ml1m_GANMF_user.index"))
    assert e["autoencoder/encoding/kernel"]["shape"] == (3706, 992)
    assert e["generator/user_embeddings"]["shape"] == (6040, 250)
    assert {k: {"shape": list(v["shape"]), "offset": v["offset"], "size": v["size"]} for k, v in e.items()} == chk["ml1m_user_entries"]


def test_writer_matches_reference_structure_and_round_trips(golden_dir, tmp_path):
    """Same names and shapes as the reference checkpoint -> an index that differs from the reference's only in the
    CRC fields of the two tensors whose data is not held as a fixture."""
    t = np.load(os.path.join(golden_dir, "kat1_checkpoint_tensors.npz"))
    rng = np.random.RandomState(0)
    tensors = {"autoencoder/decoding/bias": t["bd"], "autoencoder/decoding/kernel": rng.randn(133, 1884),
               "autoencoder/encoding/bias": t["be"], "autoencoder/encoding/kernel": rng.randn(1884, 133),
               "generator/item_embeddings": t["V"], "generator/user_embeddings": t["U"]}
    prefix = str(tmp_path / "GANMF_item")
    tb.write_bundle(prefix, tensors)
    ref = open(os.path.join(golden_dir, "kat1_GANMF_item.index"), "rb").read()
    got = open(prefix + ".index", "rb").read()
    assert len(got) == len(ref) == 354
    diff = [i for i in range(len(ref)) if ref[i] != got[i]]
    # bytes that may differ: the fixed32 CRCs of the two random kernels and the data-block trailer CRC
    allowed = set(range(0x55, 0x59)) | set(range(0x9e, 0xa2)) | set(range(0x10d, 0x111))
    assert set(diff) <= allowed, [hex(i) for i in diff]
    back = tb.read_bundle(prefix)
    for k, v in tensors.items():
        assert np.array_equal(back[k], np.asarray(v, dtype=np.float32))
    assert os.path.getsize(prefix + ".data-00000-of-00001") == 2090708


def test_many_tensors_cross_restart_interval_and_corruption(tmp_path):
    rng = np.random.RandomState(1)
    tensors = {"discriminator/layer_%d/%s" % (l, kind): rng.randn(*(shape)).astype(np.float32)
               for l in range(12) for kind, shape in (("kernel", (5, 7)), ("bias", (7,)))}
    tensors["generator/user_embeddings"] = rng.randn(9, 3).astype(np.float32)
    tensors["scalar_like"] = rng.randn(1).astype(np.float32)
    prefix = str(tmp_path / "many")
    tb.write_bundle(prefix, tensors)                      # 27 entries: two restart points
    back = tb.read_bundle(prefix)
    assert set(back) == set(tensors) and all(np.array_equal(back[k], tensors[k]) for k in tensors)
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[10] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum mismatch in tensor"):
        tb.read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[20] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="block checksum"):
        tb.read_index(prefix + ".index")
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError, match="not a tensor-bundle index"):
        tb.read_index(prefix + ".index")
