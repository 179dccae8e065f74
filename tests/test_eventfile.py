"""The dependency-free TensorBoard event-file writer: TFRecord framing with masked CRC-32C, hand-encoded
Event / Summary protobufs, read back by the module's own reader (which verifies every CRC)."""
import os
import struct

import numpy as np

import safe_grid_agents_amd as S
from safe_grid_agents_amd import eventfile as E
from oracle.gym_shim import OracleGridworldEnv


def test_crc32c_known_answers():
    assert E.crc32c(b"123456789") == 0xE3069283  # the CRC-32C check value (RFC 3720 appendix B.4)
    assert E.crc32c(b"") == 0
    assert E.crc32c(bytes(32)) == 0x8A9136AA  # 32 zero bytes (RFC 3720 test pattern)
    crc = E.crc32c(b"123456789")
    assert E.masked_crc32c(b"123456789") == ((((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def test_round_trip_of_every_call(tmp_path):
    import torch

    w = S.EventFileWriter(str(tmp_path))
    w.add_scalar("Train/returns", -62, 3)
    w.add_scalar("Train/policy_entropy", torch.tensor(1.25, requires_grad=True), 4)
    w.add_scalars("Evaluation/returns", {"avg": -10.5, "max": 7}, 2)
    w.add_text("data/seed", "7")
    w.add_histogram("network.0.0.weight", np.arange(100, dtype=np.float32) / 10.0, 9)
    clips = np.zeros((2, 3, 4, 5, 5), dtype=np.uint8)  # the reference's stacking: (clips, colour, time, H, W)
    for t in range(4):  # a pixel that moves: identical consecutive frames would be merged by the GIF encoder
        clips[0, 0, t, 2, t] = 255
    clips[1, 2, :, 1, 3] = 200
    w.add_video("Evaluation/grid_animation", clips, 6)
    w.add_scalar("negative/step", 1.0, -5)
    w.close()
    assert w.dropped_videos == 0 and os.path.basename(w.path).startswith("events.out.tfevents.")
    ev = S.read_events(w.path)
    assert ev[0] == {"step": 0, "tag": None, "kind": "file_version", "value": "brain.Event:2"}
    got = [(e["step"], e["tag"], e["kind"]) for e in ev[1:]]
    assert got == [(3, "Train/returns", "scalar"), (4, "Train/policy_entropy", "scalar"), (2, "Evaluation/returns/avg", "scalar"),
                   (2, "Evaluation/returns/max", "scalar"), (0, "data/seed/text_summary", "text"),
                   (9, "network.0.0.weight", "histogram"), (6, "Evaluation/grid_animation", "image"),
                   (-5, "negative/step", "scalar")]
    from PIL import Image
    import io

    im = ev[7]["value"]
    gif = Image.open(io.BytesIO(im["encoded"]))
    assert gif.format == "GIF" and gif.n_frames == 4 and gif.size == (im["width"], im["height"]) and im["colorspace"] == 3
    assert im["width"] == 2 * im["height"] and im["height"] >= 64  # two clips side by side, enlarged
    gif.seek(1)
    frame = np.asarray(gif.convert("RGB"))
    k = im["height"] // 5
    assert tuple(frame[2 * k + k // 2, 1 * k + k // 2]) == (255, 0, 0) and tuple(frame[k + k // 2, (5 + 3) * k + k // 2]) == (0, 0, 200)
    assert ev[1]["value"] == -62.0 and ev[2]["value"] == 1.25 and ev[3]["value"] == -10.5 and ev[5]["value"] == "7"
    h = ev[6]["value"]
    assert h["num"] == 100 and h["min"] == 0.0 and abs(h["max"] - 9.9) < 1e-6 and sum(h["bucket"]) == 100
    assert len(h["bucket"]) == len(h["bucket_limit"]) == 30 and abs(h["sum"] - 495.0) < 1e-3


def test_framing_is_tfrecord(tmp_path):
    w = S.EventFileWriter(str(tmp_path))
    w.add_scalar("x", 2.5, 1)
    w.close()
    raw = open(w.path, "rb").read()
    (n,) = struct.unpack_from("<Q", raw, 0)
    assert struct.unpack_from("<I", raw, 8)[0] == E.masked_crc32c(raw[:8])
    first = raw[12:12 + n]
    assert first[0] == 0x09 and b"brain.Event:2" in first  # field 1 (wall_time, fixed64) comes first
    # the scalar record: ... 0x2a (field 5, summary) { 0x0a (value) { 0x0a tag 'x', 0x15 simple_value 2.5f } }
    second = raw[16 + n:]
    assert b"\x0a\x01x\x15" + struct.pack("<f", 2.5) in second
    corrupted = bytearray(raw)
    corrupted[20] ^= 1
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(corrupted))
    try:
        S.read_events(str(bad))
        raise AssertionError("CRC mismatch not detected")
    except ValueError:
        pass


def test_train_writes_an_event_file_when_tensorboardx_is_absent(tmp_path):
    args = S.prepare_parser().parse_args(["-S", "4", "-E", "6", "-EE", "3", "-V", "120", "-EV", "0", "-L", str(tmp_path), "boat",
                                          "tabular-q", "-l", ".5"])
    S.train(args, env_factory=OracleGridworldEnv)
    files = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents.")]
    assert len(files) == 1
    ev = S.read_events(os.path.join(str(tmp_path), files[0]))
    tags = [e["tag"] for e in ev]
    assert tags.count("Train/returns") == 6 and "Train/epsilon" in tags and "Evaluation/returns/avg" in tags
    assert any(e["kind"] == "text" and e["tag"] == "data/seed/text_summary" and e["value"] == "4" for e in ev)
    returns = [e["value"] for e in ev if e["tag"] == "Train/returns"]
    assert all(-100 <= r <= 100 for r in returns)
