"""Eval side-channel: the ``maskrcnn/results.proto`` wire format.

``maskrcnn evaluate`` serialises its detections as a protobuf and hands the file to the COCOEval task
(``Sources/maskrcnn/EvaluateCommand.swift:103-118,203-248``; message layout from the generated
``Sources/maskrcnn/results.pb.swift:22-167,207-208,253-254,288-289,346-347,394-397,468-471,523``):

    Results   { repeated Result results = 1; }
    Result    { ImageInfo imageInfo = 1; repeated Detection detections = 2; }
    ImageInfo { string datasetId = 1; string id = 2; int32 width = 3; int32 height = 4; }
    Detection { double probability = 1; int32 classId = 2; string classLabel = 3; Rect boundingBox = 4; }
    Rect      { Origin origin = 1; Size size = 2; }
    Origin    { double x = 1; double y = 2; }        Size { double width = 1; double height = 2; }

This module writes/reads that wire format directly (proto3: default-valued scalars are omitted, like
SwiftProtobuf does), so the new engine's output can be consumed by the existing COCOEval task.  Host
tooling, no GPU involved.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np


def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field_no: int, wire: int) -> bytes:
    return _varint((field_no << 3) | wire)


def _f_double(no: int, v: float) -> bytes:
    return b"" if v == 0.0 and not np.signbit(v) else _key(no, 1) + struct.pack("<d", v)


def _f_int32(no: int, v: int) -> bytes:
    return b"" if v == 0 else _key(no, 0) + _varint(int(v))


def _f_string(no: int, s: str) -> bytes:
    b = s.encode()
    return b"" if not b else _key(no, 2) + _varint(len(b)) + b


def _f_msg(no: int, payload: bytes) -> bytes:
    return _key(no, 2) + _varint(len(payload)) + payload


@dataclass
class PBDetection:
    probability: float
    classId: int
    classLabel: str
    x: float
    y: float
    width: float
    height: float

    def encode(self) -> bytes:
        origin = _f_double(1, self.x) + _f_double(2, self.y)
        size = _f_double(1, self.width) + _f_double(2, self.height)
        rect = _f_msg(1, origin) + _f_msg(2, size)
        return (_f_double(1, self.probability) + _f_int32(2, self.classId) + _f_string(3, self.classLabel) + _f_msg(4, rect))


@dataclass
class PBResult:
    datasetId: str
    id: str
    width: int
    height: int
    detections: List[PBDetection] = field(default_factory=list)

    def encode(self) -> bytes:
        info = _f_string(1, self.datasetId) + _f_string(2, self.id) + _f_int32(3, self.width) + _f_int32(4, self.height)
        return _f_msg(1, info) + b"".join(_f_msg(2, d.encode()) for d in self.detections)


def encode_results(results: List[PBResult]) -> bytes:
    return b"".join(_f_msg(1, r.encode()) for r in results)


def detections_to_pb(detections: np.ndarray, class_label: str = "test") -> List[PBDetection]:
    """"detections" (N,6) rows (y1,x1,y2,x2,classId,score) → records exactly as
    EvaluateCommand.swift:203-248 builds them: probability > 0.7, Double arithmetic on the Float
    values, classLabel "test" (:220), masks dropped."""
    out = []
    for r in np.asarray(detections, dtype=np.float32):
        p = float(r[5])
        if p > 0.7:
            y1, x1, y2, x2 = (float(v) for v in r[:4])
            out.append(PBDetection(p, int(r[4]), class_label, x1, y1, x2 - x1, y2 - y1))
    return out


# ---- minimal reader (round-trip tests) -------------------------------------------------------------
def _read_varint(b: bytes, p: int):
    n = shift = 0
    while True:
        c = b[p]
        p += 1
        n |= (c & 0x7F) << shift
        if not c & 0x80:
            return n, p
        shift += 7


def _fields(b: bytes):
    p = 0
    while p < len(b):
        k, p = _read_varint(b, p)
        no, wire = k >> 3, k & 7
        if wire == 0:
            v, p = _read_varint(b, p)
        elif wire == 1:
            v = struct.unpack_from("<d", b, p)[0]
            p += 8
        elif wire == 2:
            n, p = _read_varint(b, p)
            v = b[p:p + n]
            p += n
        else:
            raise ValueError(f"unsupported wire type {wire}")
        yield no, v


def decode_results(data: bytes) -> List[PBResult]:
    res = []
    for no, payload in _fields(data):
        if no != 1:
            continue
        r = PBResult("", "", 0, 0)
        for n2, v in _fields(payload):
            if n2 == 1:
                for n3, w in _fields(v):
                    if n3 == 1: r.datasetId = w.decode()
                    elif n3 == 2: r.id = w.decode()
                    elif n3 == 3: r.width = int(w)
                    elif n3 == 4: r.height = int(w)
            elif n2 == 2:
                d = PBDetection(0.0, 0, "", 0.0, 0.0, 0.0, 0.0)
                for n3, w in _fields(v):
                    if n3 == 1: d.probability = w
                    elif n3 == 2: d.classId = int(w)
                    elif n3 == 3: d.classLabel = w.decode()
                    elif n3 == 4:
                        for n4, u in _fields(w):
                            for n5, z in _fields(u):
                                if n4 == 1 and n5 == 1: d.x = z
                                elif n4 == 1 and n5 == 2: d.y = z
                                elif n4 == 2 and n5 == 1: d.width = z
                                elif n4 == 2 and n5 == 2: d.height = z
                r.detections.append(d)
        res.append(r)
    return res
