# CountSignals: read-only list of integer vectors (or 2 x width integer matrices with rows
# "sense"/"antisense" when strand-specific).  Same class, slots, validity and methods as the
# reference's R/zzzCountSignals.R:27-143.

setClass("CountSignals", representation(signals = "list", ss = "logical"))

setValidity("CountSignals", function(object) {
    if (length(object@ss) != 1 || is.na(object@ss)) return("invalid ss slot")
    if (!checkList(object@signals, object@ss)) return("invalid list")
    TRUE
})

setMethod("length", "CountSignals", function(x) length(x@signals))

setMethod("width", "CountSignals", function(x) fastWidth(x@signals, x@ss))

setMethod("[", "CountSignals", function(x, i, drop=TRUE) {
    if (length(i) == 1 && drop) return(x@signals[[i]])
    new("CountSignals", signals=x@signals[i], ss=x@ss)
})

setMethod("as.list", "CountSignals", function(x) x@signals)
as.list.CountSignals <- function(x, ...) x@signals

setGeneric("alignSignals", function(x) standardGeneric("alignSignals"))
setMethod("alignSignals", "CountSignals", function(x) {
    w <- width(x)
    if (any(w != w[1])) stop("all signals must have the same length")
    simplify2array(x@signals)
})

.bs_show_counts <- function(v) {
    k <- min(length(v), 10)
    txt <- paste0(as.character(v[seq_len(k)]), collapse=" ")
    if (k < length(v)) paste(txt, "...") else txt
}

setMethod("show", "CountSignals", function(object) {
    n <- length(object)
    cat("CountSignals object with ", n, ifelse(object@ss, " strand-specific", ""),
        " signal", ifelse(n != 1, "s", ""), fill=TRUE, sep="")
    for (i in seq_len(min(5, n))) {
        el <- object[i]
        cat("[", i, "] signal of width ", ifelse(object@ss, ncol(el), length(el)), fill=TRUE, sep="")
        if (object@ss) {
            cat("sense      ", .bs_show_counts(el[1,]), sep="", fill=TRUE)
            cat("antisense  ", .bs_show_counts(el[2,]), sep="", fill=TRUE)
        } else cat(.bs_show_counts(el), fill=TRUE)
    }
    if (n > 5) cat("....", fill=TRUE)
})
