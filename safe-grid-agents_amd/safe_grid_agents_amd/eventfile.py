"""A dependency-free TensorBoard event-file writer with the tensorboardX.SummaryWriter calls the reference makes
(reference train.py:44-49 add_text; meters.py:86-106 add_scalar / add_scalars; eval.py:44-48 add_video;
value.py:131-135, policy_base.py:123-127 add_histogram), so that `track_metrics` output lands in files TensorBoard reads
when tensorboardX is not installed.

Format: a TFRecord stream -- little-endian uint64 length, masked CRC-32C of the length, the payload, masked CRC-32C of the
payload -- whose payloads are `tensorflow.Event` protobuf messages, encoded here by hand (a dozen fields):

  Event            1 wall_time (double)  2 step (int64)  3 file_version (string)  5 summary (Summary)
  Summary          1 value (repeated Value)
  Value            1 tag  2 simple_value (float)  5 histo (HistogramProto)  8 tensor (TensorProto)  9 metadata (SummaryMetadata)
  HistogramProto   1 min  2 max  3 num  4 sum  5 sum_squares  6 bucket_limit (packed doubles)  7 bucket (packed doubles)
  SummaryMetadata  1 plugin_data { 1 plugin_name  2 content }
  TensorProto      1 dtype (DT_STRING = 7)  2 tensor_shape { 2 dim { 1 size } }  8 string_val (repeated bytes)

  Image            1 height  2 width  3 colorspace  4 encoded_image_string     (Value field 4)

add_scalars writes one series per key under "<main_tag>/<key>" in the same file (tensorboardX opens one sub-run per key).
add_video encodes the clips side by side as one animated GIF with Pillow when it is importable (else the call is counted
and dropped); it takes the reference's (N, C, T, H, W) stacks (eval.py:26-28,44-48) as well as tensorboardX's (N, T, C, H, W).
"""
import os
import socket
import struct
import time

import numpy as np

# ---- CRC-32C (Castagnoli), table driven; TFRecord masks it so that a CRC of data containing CRCs stays well distributed --
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format ------------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1  # negative int64 values are encoded as their two's complement, ten bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, n):
    return _key(field, 0) + _varint(n)


def _f_double(field, x):
    return _key(field, 1) + struct.pack("<d", x)


def _f_float(field, x):
    return _key(field, 5) + struct.pack("<f", x)


def _f_bytes(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _f_packed_doubles(field, xs):
    return _f_bytes(field, struct.pack("<%dd" % len(xs), *xs))


def _event(wall_time, step=None, file_version=None, summary=None):
    msg = _f_double(1, wall_time)
    if step is not None:
        msg += _f_varint(2, int(step))
    if file_version is not None:
        msg += _f_bytes(3, file_version.encode())
    if summary is not None:
        msg += _f_bytes(5, summary)
    return msg


def _scalar_value(tag, value):
    return _f_bytes(1, _f_bytes(1, tag.encode()) + _f_float(2, float(value)))


class EventFileWriter:
    """add_scalar / add_scalars / add_text / add_histogram / add_video / flush / close on one events.out.tfevents file."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir or "runs"
        os.makedirs(self.log_dir, exist_ok=True)
        name = "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname(), os.getpid())
        self.path = os.path.join(self.log_dir, name)
        self._f = open(self.path, "wb")
        self.dropped_videos = 0
        self._unflushed = 0
        self._record(_event(time.time(), file_version="brain.Event:2"))

    def _record(self, payload):
        header = struct.pack("<Q", len(payload))
        self._f.write(header + struct.pack("<I", masked_crc32c(header)) + payload + struct.pack("<I", masked_crc32c(payload)))
        self._unflushed += 1
        if self._unflushed >= 64:  # a crashed run keeps all but its last few records
            self.flush()

    def _summary(self, value_msgs, step):
        self._record(_event(time.time(), step=step, summary=b"".join(value_msgs)))

    @staticmethod
    def _number(v):
        if hasattr(v, "detach"):  # 0-d torch tensor (PPO logs its entropy as one)
            v = v.detach()
        return float(v)

    def add_scalar(self, tag, value, step=0):
        self._summary([_scalar_value(tag, self._number(value))], step)

    def add_scalars(self, main_tag, values, step=0):
        self._summary([_scalar_value("%s/%s" % (main_tag, k), self._number(v)) for k, v in values.items()], step)

    def add_text(self, tag, text, step=0):
        tensor = _f_varint(1, 7) + _f_bytes(2, _f_bytes(2, _f_varint(1, 1))) + _f_bytes(8, str(text).encode())
        metadata = _f_bytes(1, _f_bytes(1, b"text"))
        self._summary([_f_bytes(1, _f_bytes(1, (tag + "/text_summary").encode()) + _f_bytes(8, tensor) + _f_bytes(9, metadata))], step)

    def add_histogram(self, tag, values, step=0, bins=30):
        v = np.asarray(values, dtype=np.float64).ravel()
        if v.size == 0:
            return
        counts, edges = np.histogram(v, bins=bins)
        histo = (_f_double(1, v.min()) + _f_double(2, v.max()) + _f_double(3, float(v.size)) + _f_double(4, v.sum())
                 + _f_double(5, float((v * v).sum())) + _f_packed_doubles(6, edges[1:].tolist())
                 + _f_packed_doubles(7, counts.astype(np.float64).tolist()))
        self._summary([_f_bytes(1, _f_bytes(1, tag.encode()) + _f_bytes(5, histo))], step)

    def add_video(self, tag, frames, step=0, fps=4):
        try:
            from PIL import Image
        except ImportError:
            self.dropped_videos += 1
            return
        import io

        v = np.asarray(frames)
        if v.ndim != 5:
            raise ValueError("add_video expects a 5-D array, got shape %r" % (v.shape,))
        if v.shape[1] in (1, 3) and v.shape[2] not in (1, 3):  # (N, C, T, H, W), as the reference stacks its clips
            v = np.swapaxes(v, 1, 2)
        v = np.moveaxis(v, 2, -1)  # (N, T, H, W, C)
        if v.dtype != np.uint8:
            v = np.clip(v * (255.0 if v.max() <= 1.0 else 1.0), 0, 255).astype(np.uint8)
        if v.shape[-1] == 1:
            v = np.repeat(v, 3, axis=-1)
        strip = np.concatenate(list(v), axis=2)  # clips side by side: (T, H, N * W, 3)
        scale = max(1, -(-64 // min(strip.shape[1], strip.shape[2])))  # gridworld boards are a few pixels: enlarge
        strip = strip.repeat(scale, axis=1).repeat(scale, axis=2)
        images = [Image.fromarray(f) for f in strip]
        buf = io.BytesIO()
        images[0].save(buf, format="GIF", save_all=True, append_images=images[1:], duration=int(1000 / max(fps, 1)), loop=0)
        image = (_f_varint(1, strip.shape[1]) + _f_varint(2, strip.shape[2]) + _f_varint(3, 3) + _f_bytes(4, buf.getvalue()))
        self._summary([_f_bytes(1, _f_bytes(1, tag.encode()) + _f_bytes(4, image))], step)

    def flush(self):
        self._f.flush()
        self._unflushed = 0

    def close(self):
        if not self._f.closed:
            self._f.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- a reader for tests and tooling: yields (step, tag, kind, value) and verifies every CRC ---------------------------
def _parse(buf):
    """protobuf message -> list of (field, wire, value); nested messages stay bytes."""
    out, i = [], 0
    while i < len(buf):
        key, shift = 0, 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, shift = 0, 0
            while True:
                b = buf[i]
                i += 1
                val |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
        elif wire == 1:
            val = struct.unpack_from("<d", buf, i)[0]
            i += 8
        elif wire == 5:
            val = struct.unpack_from("<f", buf, i)[0]
            i += 4
        elif wire == 2:
            n, shift = 0, 0
            while True:
                b = buf[i]
                i += 1
                n |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            val = bytes(buf[i:i + n])
            i += n
        else:
            raise ValueError("wire type %d" % wire)
        out.append((field, wire, val))
    return out


def read_events(path):
    """List of dicts {step, tag, kind, value} for every summary value in an event file; raises on a CRC mismatch."""
    data = open(path, "rb").read()
    events, i = [], 0
    while i < len(data):
        header = data[i:i + 8]
        (n,) = struct.unpack("<Q", header)
        if struct.unpack_from("<I", data, i + 8)[0] != masked_crc32c(header):
            raise ValueError("length CRC mismatch at byte %d" % i)
        payload = data[i + 12:i + 12 + n]
        if struct.unpack_from("<I", data, i + 12 + n)[0] != masked_crc32c(payload):
            raise ValueError("payload CRC mismatch at byte %d" % i)
        i += 16 + n
        fields = _parse(payload)
        step = next((v for f, w, v in fields if f == 2), 0)
        if step >= 1 << 63:
            step -= 1 << 64
        for f, w, v in fields:
            if f == 3:
                events.append({"step": 0, "tag": None, "kind": "file_version", "value": v.decode()})
            if f != 5:
                continue
            for f2, w2, value_msg in _parse(v):
                vf = _parse(value_msg)
                tag = next(x for g, _, x in vf if g == 1).decode()
                for g, w3, x in vf:
                    if g == 2:
                        events.append({"step": step, "tag": tag, "kind": "scalar", "value": x})
                    elif g == 5:
                        h = {g2: x2 for g2, _, x2 in _parse(x)}
                        events.append({"step": step, "tag": tag, "kind": "histogram",
                                       "value": {"min": h[1], "max": h[2], "num": h[3], "sum": h[4], "sum_squares": h[5],
                                                 "bucket_limit": list(struct.unpack("<%dd" % (len(h[6]) // 8), h[6])),
                                                 "bucket": list(struct.unpack("<%dd" % (len(h[7]) // 8), h[7]))}})
                    elif g == 4:
                        im = {g2: x2 for g2, _, x2 in _parse(x)}
                        events.append({"step": step, "tag": tag, "kind": "image",
                                       "value": {"height": im[1], "width": im[2], "colorspace": im[3], "encoded": im[4]}})
                    elif g == 8:
                        t = {g2: x2 for g2, _, x2 in _parse(x)}
                        events.append({"step": step, "tag": tag, "kind": "text", "value": t[8].decode()})
    return events
