# bamCount / bamProfile / bamCoverage: same generics, signatures, defaults, argument
# normalisation, warnings, errors and return shapes as the reference's R/wrappers.R:75-184.
# Everything below the three methods happens in the native library.

# SAM flag mask required of a read: 0x2 | 0x40 (first read of a proper pair) unless the
# pairing is ignored (reference: flagMask, R/wrappers.R:76-81)
.bs_required_flag <- function(pe) if (identical(pe, "ignore")) 0L else 66L

# template-length filter handed to the native code (reference: tlenFilter, R/wrappers.R:84-98)
.bs_tlen_filter <- function(tf, pe) {
    if (identical(pe, "ignore")) return(integer())
    if (is.null(tf)) return(c(0, 1000))
    if (length(tf) != 2 || tf[1] < 0 || tf[2] < 0)
        stop("tlenFilter must be NULL or vector of 2 positive integers")
    if (tf[1] > tf[2])
        stop("tlenFilter[1] must be smaller or equal to tlenFilter[2]")
    tf
}

.bs_sentences <- c(
    "tHaT'S tHa fAStEsT pIlE-uP bAm iN tHe SoUth!!!",
    "yOu cAn'T pIlE-Up FaStEr!!!",
    "I'M gOnNa cHaSe'em and PiLe'em aLl up!!!",
    "fOr brOoMmHiLdA!!!",
    "tHe lEgEnD said, hE cOuLd PiLe uP fAsTeR thAn LiGht",
    "I gEt gOoSeBuMPs wHen I seE yOu pilEuPpiNg...")
.bs_announce <- function(path) message("Processing ", path, ": ", sample(.bs_sentences, 1))

setGeneric("bamCount", function(bampath, gr, ...) standardGeneric("bamCount"))
setMethod("bamCount", c("character", "GenomicRanges"),
    function(bampath, gr, mapqual=0, shift=0, ss=FALSE,
             paired.end=c("ignore", "filter", "midpoint"),
             tlenFilter=NULL, filteredFlag=-1, verbose=TRUE) {
        if (verbose) .bs_announce(bampath)
        pe <- match.arg(paired.end)
        res <- pileup_core(path.expand(bampath), gr, .bs_tlen_filter(tlenFilter, pe), mapqual,
                           -1, shift, ss, .bs_required_flag(pe), filteredFlag, pe == "midpoint")
        res[[1]]
    })

setGeneric("bamProfile", function(bampath, gr, ...) standardGeneric("bamProfile"))
setMethod("bamProfile", c("character", "GenomicRanges"),
    function(bampath, gr, binsize=1, mapqual=0, shift=0, ss=FALSE,
             paired.end=c("ignore", "filter", "midpoint"),
             tlenFilter=NULL, filteredFlag=-1, verbose=TRUE) {
        if (verbose) .bs_announce(bampath)
        if (binsize < 1) stop("provide a binsize greater or equal to 1")
        if (binsize > 1 && any((width(gr) %% binsize) != 0))
            warning("some ranges' widths are not a multiple of the selected
             binsize, some bins will correspond to less than binsize basepairs")
        pe <- match.arg(paired.end)
        res <- pileup_core(path.expand(bampath), gr, .bs_tlen_filter(tlenFilter, pe), mapqual,
                           binsize, shift, ss, .bs_required_flag(pe), filteredFlag,
                           pe == "midpoint")
        new("CountSignals", signals=res, ss=ss)
    })

setGeneric("bamCoverage", function(bampath, gr, ...) standardGeneric("bamCoverage"))
setMethod("bamCoverage", c("character", "GenomicRanges"),
    function(bampath, gr, mapqual=0, paired.end=c("ignore", "extend"),
             tlenFilter=NULL, filteredFlag=-1, verbose=TRUE) {
        if (verbose) .bs_announce(bampath)
        pe <- match.arg(paired.end)
        res <- coverage_core(path.expand(bampath), gr, .bs_tlen_filter(tlenFilter, pe), mapqual,
                             .bs_required_flag(pe), filteredFlag, pe == "extend")
        new("CountSignals", signals=res, ss=FALSE)
    })
