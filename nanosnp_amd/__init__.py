"""nanosnp_amd -- MI355X (gfx950) implementation of NanoSNP's candidate-site inference hot path.

The compute path is the HIP library behind include/nanosnp.h (``nanosnp_amd/_lib.py`` binds it);
this package holds only the host-side mirror of the reference's interface for that path:

    pileup_model.LSTMNetwork     PileupModel/model.py LSTMNetwork (predict only)
    haplotype_model.LSTMNetwork  HaplotypeModel/model_dev.py LSTMNetwork (predict only)
    cat_model.CatModel           HaplotypeModel/model.py CatModel, the legacy path (predict only)
    pipeline                     mpileup text -> column encode -> windows -> PileupModel -> pileup.vcf in one pass
                                 (dna_sv_tensor make_candidate_snp_tensor + make_predict_data + predict.py)
                                 and .pd.bin window files -> pinned staging / H2D / PileupModel -> pileup.vcf, streamed
                                 (PileupModel/predict.py + dataset.py PredictDataset behind a DataLoader)
    hap_pipeline                 haplotype site files -> pinned staging / H2D / features + HaplotypeModel -> haplotype.csv, streamed
                                 (HaplotypeModel/predict_dev.py + dataset_dev.py TestDataset behind a DataLoader)
    predict                      PileupModel/predict.py / HaplotypeModel/predict_dev.py loops on arrays
    readmatrix                   stage-4 read x position matrices from a pysam-like alignment file (create_pileup_haplotype.py)
    merge                        stage-4 group selection and the final merge (select_hetesnp_homosnp.py, scripts/merge.py)
    sitefile                     flat binary containers in place of the HDF5 bins
    dist                         static site sharding over the GPUs of a node + result gather
    host                         native readers / synthetic generators (libnanosnp_host.so)

There is no CPU fallback: without the built extension and a gfx950 device every compute entry
point raises.  The CPU restatement under ``oracle/`` is test infrastructure only.
"""
__version__ = "0.5.0"
