"""Entry script: `python compress.py in.png out.bin [--block_size ...]` (the reference's compress.py; same flags)."""
from cli import compress, compress_parser as build_parser, main_compress, quantization_from_args  # noqa: F401

if __name__ == "__main__":
    main_compress()
