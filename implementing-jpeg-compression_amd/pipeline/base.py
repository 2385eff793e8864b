"""The plugin boundary of the codec (behaviour of the reference's pipeline/base.py).

A stage is a class derived from ``AlgorithmStep`` that defines a numeric ``step_index`` in its own body;
defining it is all it takes to take part: the metaclass files the class into the module-level
``step_classes`` list, which stays ordered by ``step_index`` (classes with equal indices keep the order
in which they were defined -- defining a second class with an index that is already taken ADDS a stage,
it does not replace one).  ``compress_band`` walks the list forwards, ``decompress_band`` backwards.
"""
import bisect

from util import padded_size, split_into_blocks

step_classes = []


class IndexOutOfOrderError(Exception):
    pass


class MissingStepIndexError(Exception):
    pass


class Meta(type):
    """Metaclass of AlgorithmStep: registration happens when the class statement finishes."""

    def __init__(cls, name, bases, namespace):
        super().__init__(name, bases, namespace)
        if name == "AlgorithmStep":
            return
        if "step_index" not in namespace:        # must be stated by the class itself, not inherited
            raise MissingStepIndexError('Class {} has not defined "{}" class attribute'.format(name, "step_index"))
        bisect.insort_right(step_classes, cls, key=lambda c: c.step_index)


class AlgorithmStep(metaclass=Meta):
    """One reversible stage: ``execute`` on the way in, ``invert`` on the way out; constructed per call
    with the band's Configuration and otherwise stateless."""

    def __init__(self, config):
        self._config = config

    def execute(self, array):
        raise NotImplementedError

    def invert(self, array):
        raise NotImplementedError

    def calculate_padding(self, factor):
        """(rows, cols) by which the configured image grows when padded to a multiple of ``factor``."""
        return tuple(padded_size(v, factor) - v for v in (self._config.height, self._config.width))

    def blocks(self, a, block_size):
        """(block, y, x) for every block in the codec's order: rows of blocks top to bottom, left to right."""
        tiles = split_into_blocks(a, block_size)
        for y, x in ((y, x) for y in range(a.shape[0] // block_size) for x in range(a.shape[1] // block_size)):
            yield tiles[y, x], y, x

    def apply_blockwise(self, a, transformation, block_size, res):
        """Writes ``transformation(block)`` over every block's place in ``res``."""
        for block, y, x in self.blocks(a, block_size):
            res[y * block_size:(y + 1) * block_size, x * block_size:(x + 1) * block_size] = transformation(block)
