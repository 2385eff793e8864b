"""CLI: decode a file written by compress.py back to an image (reference: decompress.py)."""
import argparse

from pipeline import Jpeg


def decompress(input_path, output_path):
    with open(input_path, "rb") as f:
        image = Jpeg.decompress(f.read())
    image.convert("RGB").save(output_path)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Decode a stream written by compress.py and save it as an image")
    parser.add_argument("infile", type=str, help="compressed stream to read")
    parser.add_argument("outfile", type=str, help="image file to write (format from the extension)")
    args = parser.parse_args()
    decompress(args.infile, args.outfile)
