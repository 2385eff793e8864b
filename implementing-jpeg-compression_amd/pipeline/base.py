"""Step registry and base class -- the reference's plugin API (reference: pipeline/base.py).

Every concrete step is a class with a ``step_index`` class attribute; creating the class
registers it in the module-level ``step_classes`` list, kept sorted by ``step_index``
(pipeline/base.py:7-31).  ``compress_band`` / ``decompress_band`` walk that list.
"""
from util import padded_size, split_into_blocks

step_classes = []


class IndexOutOfOrderError(Exception):
    pass


class MissingStepIndexError(Exception):
    pass


class Meta(type):
    """Registers every subclass of AlgorithmStep (pipeline/base.py:7-31)."""

    @staticmethod
    def validate_index(cls, name, class_dict):
        if "step_index" not in class_dict:
            raise MissingStepIndexError(
                'Class {} has not defined "{}" class attribute'.format(name, "step_index"))

    @staticmethod
    def sort_classes():
        step_classes.sort(key=lambda c: c.step_index)

    def __new__(meta, name, bases, class_dict):
        cls = super().__new__(meta, name, bases, class_dict)
        if name != "AlgorithmStep":
            Meta.validate_index(cls, name, class_dict)
            step_classes.append(cls)
            Meta.sort_classes()
        return cls


class AlgorithmStep(metaclass=Meta):
    """One reversible stage of the codec: ``execute`` on the way in, ``invert`` on the way out."""

    def __init__(self, config):
        self._config = config

    def execute(self, array):
        raise NotImplementedError

    def invert(self, array):
        raise NotImplementedError

    def calculate_padding(self, factor):
        """(rows, cols) added when the configured image is padded to a multiple of ``factor``."""
        h, w = self._config.height, self._config.width
        return padded_size(h, factor) - h, padded_size(w, factor) - w

    def blocks(self, a, block_size):
        """Yield (block, y, x) in the codec's block order: y outer, x inner (pipeline/base.py:58-66)."""
        tiles = split_into_blocks(a, block_size)
        for y in range(a.shape[0] // block_size):
            for x in range(a.shape[1] // block_size):
                yield tiles[y, x], y, x

    def apply_blockwise(self, a, transformation, block_size, res):
        """res[block] = transformation(block) for every block (pipeline/base.py:68-72)."""
        for block, y, x in self.blocks(a, block_size):
            res[y * block_size:(y + 1) * block_size, x * block_size:(x + 1) * block_size] = transformation(block)
