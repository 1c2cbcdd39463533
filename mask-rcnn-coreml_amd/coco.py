"""COCO ``instances_*.json`` reader — the data format on the input side of the evaluate harness.

Mirrors ``Sources/maskrcnn/COCO.swift``: ``COCO(url:)`` decodes ``info`` / ``images`` / ``annotations`` (:52-58; snake_case keys,
:55), ``index`` groups the annotations by ``image_id`` in file order (:9-24), and ``makeImageIterator(limit:sortById:)`` yields
``(image, annotations)`` over the images — optionally sorted by id — cut to the first ``limit`` (:60-78; the evaluate command
uses ``limit: 5, sortById: true``, ``EvaluateCommand.swift:165``).  Like the reference, a ``limit`` larger than the image count is
an error (Swift traps on the out-of-range slice; here: ValueError).  Only the fields the reference declares are kept
(``COCOImage``: id, file_name, width, height :93-99; ``COCOAnnotation``: id, image_id, category_id, bbox :101-107).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple


@dataclass(frozen=True)
class COCOImage:
    id: int
    fileName: str
    width: int
    height: int


@dataclass(frozen=True)
class COCOAnnotation:
    id: int
    imageId: int
    categoryId: int
    bbox: Tuple[float, ...]


class COCO:
    def __init__(self, path: str):
        with open(path, "r", encoding="utf-8") as f:
            doc = json.load(f)
        for key in ("info", "images", "annotations"):           # COCOInstances :79-83: all three are required
            if key not in doc:
                raise ValueError(f"{path}: missing key '{key}'")
        self.info = doc["info"]
        try:
            self.images: List[COCOImage] = [COCOImage(int(i["id"]), str(i["file_name"]), int(i["width"]), int(i["height"])) for i in doc["images"]]
            self.annotations: List[COCOAnnotation] = [
                COCOAnnotation(int(a["id"]), int(a["image_id"]), int(a["category_id"]), tuple(float(v) for v in a["bbox"]))
                for a in doc["annotations"]]
        except KeyError as e:
            raise ValueError(f"{path}: record without required field {e}") from None
        self._index: Optional[Dict[int, List[COCOAnnotation]]] = None

    @property
    def index(self) -> Dict[int, List[COCOAnnotation]]:
        """annotationsByImageIds (:9-24), built lazily like the reference's ``lazy var``."""
        if self._index is None:
            idx: Dict[int, List[COCOAnnotation]] = {}
            for a in self.annotations:
                idx.setdefault(a.imageId, []).append(a)
            self._index = idx
        return self._index

    def makeImageIterator(self, limit: Optional[int] = None, sortById: bool = False) -> Iterator[Tuple[COCOImage, List[COCOAnnotation]]]:
        images = sorted(self.images, key=lambda i: i.id) if sortById else list(self.images)     # stable, like the test data needs
        if limit is not None:
            if limit < 0 or limit > len(images):
                raise ValueError(f"limit {limit} outside 0..{len(images)} (the reference slices images[0..<limit])")
            images = images[:limit]
        idx = self.index
        for im in images:
            yield im, idx.get(im.id, [])
