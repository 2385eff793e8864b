"""Command lines of the codec: the flag surface of the reference's compress.py / decompress.py
(compress.py:27-62, decompress.py:13-25) in one place; `compress.py` and `decompress.py` are the entry scripts."""
import argparse

from pipeline import Configuration, Jpeg, QuantizationMethod

# name, type, default, help -- the reference's optional flags and their defaults
_COMPRESS_FLAGS = (
    ("--block_size", int, 4, "edge of the square tiles averaged by the sub-sampling stage (1 = off)"),
    ("--dct_size", int, 8, "edge of the transform blocks"),
    ("--transform", str, "DCT", "block transform: DCT or DFT"),
    ("--quantization", str, "qtable", "quantiser: none, discard, divide or qtable"),
    ("--qkeep", int, 2, "with --quantization discard: keep the top-left qkeep x qkeep coefficients of a block"),
    ("--qdivisor", int, 40, "with --quantization divide: the divisor applied to every coefficient"),
)


def compress_parser():
    p = argparse.ArgumentParser(description="Compress an image file with the JPEG-like block codec")
    p.add_argument("infile", type=str, help="image to read (any format Pillow opens)")
    p.add_argument("outfile", type=str, help="where to write the compressed stream")
    for flag, kind, default, text in _COMPRESS_FLAGS:
        p.add_argument(flag, action="store", type=kind, default=default, help=text)
    return p


def decompress_parser():
    p = argparse.ArgumentParser(description="Decode a stream written by compress.py and save it as an image")
    p.add_argument("infile", type=str, help="compressed stream to read")
    p.add_argument("outfile", type=str, help="image file to write (format from the extension)")
    return p


def quantization_from_args(args):
    """Anything but discard / divide / qtable means 'no quantisation object' (Configuration then rounds only)."""
    builders = {"discard": lambda: QuantizationMethod("discard", keep=args.qkeep),
                "divide": lambda: QuantizationMethod("divide", divisor=args.qdivisor),
                "qtable": lambda: QuantizationMethod("qtable")}
    make = builders.get(args.quantization)
    return make() if make else None


def compress(input_fname, output_fname, block_size=4, dct_size=8, transform="DCT", quantization=None):
    from PIL import Image
    image = Image.open(input_fname).convert("YCbCr")
    config = Configuration(width=image.width, height=image.height, block_size=block_size, dct_size=dct_size,
                           transform=transform, quantization=quantization)
    with open(output_fname, "wb") as sink:
        sink.write(Jpeg(config).compress(image))


def decompress(input_path, output_path):
    with open(input_path, "rb") as source:
        Jpeg.decompress(source.read()).convert("RGB").save(output_path)


def main_compress(argv=None):
    args = compress_parser().parse_args(argv)
    compress(args.infile, args.outfile, block_size=args.block_size, dct_size=args.dct_size, transform=args.transform,
             quantization=quantization_from_args(args))


def main_decompress(argv=None):
    args = decompress_parser().parse_args(argv)
    decompress(args.infile, args.outfile)
