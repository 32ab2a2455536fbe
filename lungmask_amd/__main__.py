"""`python -m lungmask_amd INPUT OUTPUT [...]` -- the reference's command line (lungmask/__main__.py:20-144)
on the MI355X engine.  Same flags; thin by design (all I/O, off the hot path):

* `.npy` / `.npz`, NIfTI-1 (`.nii`, `.nii.gz`), MetaImage (`.mha`, `.mhd`) and uncompressed DICOM (files and series
  folders) are read and written by `volume_io.py` without any imaging dependency -- DICOM output as one multi-frame file
  with the carried-over study / patient tags of __main__.py:125-141 (`--removemetadata` drops them);
* every other format (NRRD, compressed DICOM, ...) goes through SimpleITK like the reference (imported lazily; this image
  does not ship it);
* `--cpu` is an error by default (there is no CPU path in this engine); with LUNGMASK_AMD_ALLOW_CPU_FLAG=1 it is accepted, warned about, and the work runs on the MI355X.
"""
import argparse
import os
import sys

from .logger import logger
from .mask import LMInferer

VERSION = "0.2.20+mi355x"

# utils.py:17-30
DICOM_METADATA_TO_KEEP = ("0008|0020", "0008|0030", "0008|0050", "0008|0090", "0008|1030", "0010|0010", "0010|0020",
                          "0010|0030", "0010|0040", "0018|5100", "0020|000d", "0020|0010")


def path(string):  # __main__.py:13-17
    if os.path.exists(string):
        return string
    sys.exit(f"File not found: {string}")


def build_parser():
    p = argparse.ArgumentParser(prog="lungmask_amd", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("input", metavar="input", type=path, help="Path to the input image, can be a folder for dicoms")
    p.add_argument("output", metavar="output", type=str, help="Filepath for output lungmask")
    p.add_argument("--modelname", help="spcifies the trained model, Default: R231", type=str,
                   choices=["R231", "LTRCLobes", "LTRCLobes_R231", "R231CovidWeb"], default="R231")
    p.add_argument("--modelpath", help="spcifies the path to the trained model", default=None)
    p.add_argument("--cpu", help="Force using the CPU (not available in this engine: an error unless LUNGMASK_AMD_ALLOW_CPU_FLAG=1, then ignored with a warning)", action="store_true")
    p.add_argument("--nopostprocess", help="Deactivates postprocessing (removal of unconnected components and hole filling)", action="store_true")
    p.add_argument("--batchsize", type=int, help="Number of slices processed simultaneously.", default=20)
    p.add_argument("--noprogress", action="store_true", help="If set, no tqdm progress bar will be shown")
    p.add_argument("--version", help="Shows the current version of lungmask", action="version", version=VERSION)
    p.add_argument("--removemetadata", action="store_true", help="Do not keep study/patient related metadata of the input, if any.")
    return p


def main(argv=None):
    from . import volume_io

    args = build_parser().parse_args(sys.argv[1:] if argv is None else argv)
    keepmetadata = not args.removemetadata
    logger.info("Load model")
    image = volume_io.load_input_image(args.input)  # utils.load_input_image (utils.py:233-269)
    logger.info("Infer lungmask")
    if args.modelname == "LTRCLobes_R231":
        assert args.modelpath is None, "Modelpath can not be specified for LTRCLobes_R231 mode"
        inferer = LMInferer(modelname="LTRCLobes", force_cpu=args.cpu, fillmodel="R231", batch_size=args.batchsize,
                            volume_postprocessing=not args.nopostprocess, tqdm_disable=args.noprogress)
    else:
        inferer = LMInferer(modelname=args.modelname, modelpath=args.modelpath, force_cpu=args.cpu, batch_size=args.batchsize,
                            volume_postprocessing=not args.nopostprocess, tqdm_disable=args.noprogress)
    result = inferer.apply(image)
    logger.info(f"Save result to: {args.output}")
    keep = None
    if keepmetadata:  # __main__.py:125-141 (only formats that store tags use them)
        keep = {k: v for k, v in image.meta.items() if k in DICOM_METADATA_TO_KEEP}
        keep.update({"0008|103e": "Created with lungmask", "0028|1050": "1", "0028|1051": "2"})
    volume_io.save_image(args.output, image.like(result), keep)
    return 0


if __name__ == "__main__":
    sys.exit(main())
