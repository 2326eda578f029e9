"""Region-feature extractor of GRIT (Deformable-DETR style decoder).  Detector pre-training pieces of the
reference (models/detection/{detector,heads,od_losses}.py) are out of scope of the captioning hot path."""
