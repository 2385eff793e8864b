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
    p = argparse.ArgumentParser(description="Given an image, compress it using JPEG algorithm")
    p.add_argument("infile", type=str, help="a path to the file to compress")
    p.add_argument("outfile", type=str, help="a destination path")
    p.add_argument("--block_size", action="store", type=int, default=4, help="size of sub-sampling block")
    p.add_argument("--dct_size", action="store", type=int, default=8, help="size of block for DCT transform")
    p.add_argument("--transform", action="store", type=str, default="DCT",
                   help="type of discrete transform (DCT vs DFT)")
    p.add_argument("--quantization", action="store", type=str, default="qtable",
                   help="type of quantization to use: on of none, discard, divide, qtable ")
    p.add_argument("--qkeep", action="store", type=int, default=2,
                   help="specifies how many coefficients to keep along both axes,"
                        "applied only if quantization is set to \"discard\"")
    p.add_argument("--qdivisor", action="store", type=int, default=40,
                   help="specifies an integer used to divide coefficients by,"
                        "applied only if quantization is set to \"divide\"")
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
