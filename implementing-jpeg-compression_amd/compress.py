"""CLI: compress an image with the JPEG-like codec (reference: compress.py; same flags)."""
import argparse

from pipeline import Configuration, Jpeg, QuantizationMethod


def compress(input_fname, output_fname, block_size=4, dct_size=8, transform="DCT", quantization=None):
    from PIL import Image
    im = Image.open(input_fname).convert("YCbCr")
    config = Configuration(width=im.width, height=im.height, block_size=block_size, dct_size=dct_size,
                           transform=transform, quantization=quantization)
    with open(output_fname, "wb") as f:
        f.write(Jpeg(config).compress(im))


def build_parser():
    p = argparse.ArgumentParser(description="Compress an image file with the JPEG-like block codec")
    p.add_argument("infile", type=str, help="image to read (any format Pillow opens)")
    p.add_argument("outfile", type=str, help="where to write the compressed stream")
    p.add_argument("--block_size", action="store", type=int, default=4, help="edge of the square tiles averaged by the sub-sampling stage (1 = off)")
    p.add_argument("--dct_size", action="store", type=int, default=8, help="edge of the transform blocks")
    p.add_argument("--transform", action="store", type=str, default="DCT",
                   help="block transform: DCT or DFT")
    p.add_argument("--quantization", action="store", type=str, default="qtable",
                   help="quantiser: none, discard, divide or qtable")
    p.add_argument("--qkeep", action="store", type=int, default=2,
                   help="with --quantization discard: keep the top-left qkeep x qkeep coefficients of a block")
    p.add_argument("--qdivisor", action="store", type=int, default=40,
                   help="with --quantization divide: the divisor applied to every coefficient")
    return p


def quantization_from_args(args):
    """compress.py:53-60: anything but discard/divide/qtable means 'no quantisation object'."""
    if args.quantization == "discard":
        return QuantizationMethod("discard", keep=args.qkeep)
    if args.quantization == "divide":
        return QuantizationMethod("divide", divisor=args.qdivisor)
    if args.quantization == "qtable":
        return QuantizationMethod("qtable")
    return None


if __name__ == "__main__":
    args = build_parser().parse_args()
    compress(args.infile, args.outfile, block_size=args.block_size, dct_size=args.dct_size,
             transform=args.transform, quantization=quantization_from_args(args))
